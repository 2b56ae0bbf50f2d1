import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
import limg_amd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
g = limg_amd.LimgHip(0)
img = g.synth_device("photo_noise", 1024, 1024, seed=1)
planes = g.alloc_planes_device(1024, 1024)
g.encode3d_device(img, True, planes)
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", dist.get_backend(), float(t.item()), int(planes["pDecoded"].sum().item()) != 0)
g.close()
dist.destroy_process_group()
