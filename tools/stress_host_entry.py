"""Stress: host entry vs device entry vs repeated device entries at 4096^2 (looking for a rare mismatch)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, limg_amd
from limg_amd import PLANES
g = limg_amd.LimgHip(0)
W = 4096
bad = 0
small = torch.zeros((64, 512), dtype=torch.int32, device="cuda")
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    if it % 3 == 0:  # a small encode in between, like the test order that failed
        g.encode3d_device(small, True, g.alloc_planes_device(512, 64))
    d_img = g.synth_device("random_gradient", W, W, seed=1 + (it % 2))
    img = d_img.cpu().numpy().view(np.uint32)
    got = g.encode3d(img, True)
    planes = g.alloc_planes_device(W, W)
    g.encode3d_device(d_img, True, planes)
    torch.cuda.synchronize()
    for k in PLANES:
        dev = planes[k].cpu().numpy()
        dev = dev.view(np.uint32) if dev.dtype == np.int32 else dev
        if not np.array_equal(got[k], dev):
            ys, xs = np.nonzero(got[k] != dev)
            print("iter", it, "plane", k, "mismatches", ys.size, "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), flush=True)
            bad += 1
            if k == "pShiftABCX":
                from oracle.bind import Oracle
                orc = Oracle()
                y0 = int(ys.min()) // 8 * 8
                want = orc.encode3d(np.ascontiguousarray(img[y0:y0 + 8]), True)["pShiftABCX"]
                x = int(xs.min())
                print("   host-entry %08x  device-entry %08x  oracle %08x" % (got[k][y0, x], dev[y0, x], want[0, x]), flush=True)
    try:
        g.check()
    except Exception as e:
        print("iter", it, "check:", e, flush=True)
print("done, bad planes:", bad)
