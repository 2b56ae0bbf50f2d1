"""Multi-GPU sharding of the encode path.  One process per GPU, torch.distributed (backend "nccl" == RCCL
over xGMI on the GPU node, "gloo" in the CPU tests).  Blocks are independent except for the dither chain, so:

  * batch of images (BASELINE configs[3]):   image i -> rank i % world; nothing is exchanged on the data path.
  * one big image (BASELINE configs[4]):     `world` contiguous strips of whole block rows with the reference's own
    strip rule (src/limg.cpp:2114-2134 with thread_count == world, i.e. a pool of world/4 threads when world % 4 == 0):
    every rank restarts the dither chain at the seed, exactly like the reference's strips -- again no exchange.
  * the only collective is the optional reassembly of the planes on rank 0 (`gather_planes`): one variable-size gather
    per plane; on xGMI each peer->root transfer rides one link, so this is reported separately from the encode.

The encoder itself is passed in as a callable (the HIP path on the GPU box, the CPU oracle in the gloo tests), so the
sharding logic is identical in both."""
import numpy as np

PLANES32 = ("pDecoded", "pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax")
PLANES8 = ("pFactorsA", "pFactorsB", "pFactorsC")


def strip_rows(size_y, world):
    """Row range [y0, y1) of every rank: the reference's partition with thread_count == world (strip-restart semantics).
    Returns None when the rule degenerates (fewer block rows than ranks): the caller should then keep the image on one rank."""
    y_range = ((size_y // 8) // world) * 8
    if y_range == 0:
        return None
    out = []
    y = 0
    for r in range(world):
        y1 = size_y if r == world - 1 else y + y_range
        out.append((y, y1))
        y += y_range
    return out


def equivalent_pool_threads(world):
    """pool size T for which the reference makes exactly `world` strips (T * 4 == world), or None."""
    return world // 4 if world % 4 == 0 and world >= 4 else None


def batch_assignment(n_images, world, rank):
    return [i for i in range(n_images) if i % world == rank]


def encode_strip_sharded(encode, img_strip, has_alpha, **kw):
    """Every rank encodes its own strip with a fresh dither chain (pool_threads = 0 on the strip)."""
    return encode(img_strip, has_alpha, **kw)


def gather_planes(planes, rows, width, dist, dst=0):
    """Reassemble per-rank plane strips (dict name -> 2-D tensor) on rank `dst`.  `rows` = list of (y0, y1) per rank."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    out = {}
    for name, t in planes.items():
        if rank == dst:
            parts = [torch.empty((rows[r][1] - rows[r][0], width), dtype=t.dtype, device=t.device) for r in range(world)]
        else:
            parts = None
        # variable-size gather as grouped point-to-point (each peer -> root rides one xGMI link)
        if rank == dst:
            parts[dst].copy_(t)
            reqs = [dist.irecv(parts[r], src=r) for r in range(world) if r != dst]
            for q in reqs:
                q.wait()
            out[name] = torch.cat(parts, dim=0)
        else:
            dist.send(t.contiguous(), dst=dst)
    return out if rank == dst else None


def gather_bytes(buf, nbytes, dist, dst=0):
    """Variable-size gather of one byte tensor per rank (the compact LMG3 stream of a rank's strip / image) to rank `dst`: sizes first
    (one all_gather of an int64 per rank), then grouped point-to-point transfers of exactly the used bytes.  Returns the list of tensors on `dst`."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = buf.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([int(nbytes)], dtype=torch.int64, device=dev))
    sizes = [int(t.item()) for t in sizes]
    if rank == dst:
        parts = [buf[:sizes[r]] if r == dst else torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(world)]
        reqs = [dist.irecv(parts[r], src=r) for r in range(world) if r != dst]
        for q in reqs:
            q.wait()
        return parts
    dist.send(buf[:sizes[rank]].contiguous(), dst=dst)
    return None
