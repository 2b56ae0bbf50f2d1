"""Multi-GPU sharding of the encode path.  One process per GPU, torch.distributed (backend "nccl" == RCCL
over xGMI on the GPU node, "gloo" in the CPU tests).  Blocks are independent except for the dither chain, so:

  * batch of images (BASELINE configs[3]):   image i -> rank i % world; nothing is exchanged on the data path.
  * one big image (BASELINE configs[4]):     `world` contiguous strips of whole block rows with the reference's own
    strip rule (src/limg.cpp:2114-2134 with thread_count == world, i.e. a pool of world/4 threads when world % 4 == 0):
    every rank restarts the dither chain at the seed, exactly like the reference's strips -- again no exchange.
  * the only collective is the optional reassembly on rank 0; on xGMI each peer->root transfer rides one link, so it is reported
    separately from the encode.  The product path for it is the C ABI (`limg_hip_gather_stream`: RCCL behind liblimg_hip.so, only the
    compact streams cross the links) -- `gather_streams` below is a thin caller of it; on a gloo group (CPU tests, one-card rehearsals)
    the same offsets (`limg_hip_host_gather_offsets`) are filled by torch point-to-point transfers instead.
  * one chain through all strips (== the reference with pThreadPool == nullptr) is `limg_hip_encode3d_single_chain_device`
    (`LimgHip.encode3d_single_chain_device`): an 8-byte all-gather between the E and the F step, also inside the library.

`gather_planes` (the 35 B/px reassembly) stays a torch.distributed helper: it is a rehearsal / comparison path, not what the north-star ships."""
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


def gather_streams(g, stream, nbytes, dist, dst=0, out=None):
    """Variable-size gather of every rank's LMG3 stream on rank `dst`.  Returns (buffer, offsets[world + 1]) on `dst` (piece r = buffer[offsets[r] : offsets[r] + size_r],
    16-byte aligned, ready for decode_stream_device), None elsewhere.  RCCL group: the C ABI does all of it (the context's own communicator, created on first
    use from 128 bytes broadcast over `dist`); gloo group: same layout, torch point-to-point."""
    import torch
    from . import host_gather_offsets
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "nccl":
        if getattr(g, "comm_world", 1) != world or not getattr(g, "_comm_ready", False):
            g.comm_init_from_torch(dist)
            g._comm_ready = True
        return g.gather_stream(stream, nbytes, root=dst, out=out)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([int(nbytes)], dtype=torch.int64))
    sizes = np.array([int(t.item()) for t in sizes], dtype=np.uint64)
    offs = host_gather_offsets(sizes)
    mine = stream[:int(nbytes)].cpu().contiguous()
    if rank != dst:
        if nbytes:
            dist.send(mine, dst=dst)
        return None
    buf = torch.zeros(int(offs[world]), dtype=torch.uint8)
    reqs = []
    for r in range(world):
        piece = buf[int(offs[r]): int(offs[r]) + int(sizes[r])]
        if r == dst:
            piece.copy_(mine)
        elif sizes[r]:
            reqs.append(dist.irecv(piece, src=r))
    for q in reqs:
        q.wait()
    if out is not None:
        out[:buf.numel()].copy_(buf)
        buf = out
    return buf, offs


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


def sum64_device(t):
    """[sum e_i, sum (i + 1) e_i] over the tensor's elements (uint32 words / bytes) as unsigned 64-bit wrap-around sums: the parallel checksum
    tools/make_golden_fullsize.py records per strip next to the FNV hashes (computed there from the real reference's planes with numpy)."""
    import torch
    v = t.reshape(-1)
    if v.dtype == torch.int32:
        v = v.to(torch.int64) & 0xFFFFFFFF
    else:
        v = v.to(torch.int64)
    s1 = int(v.sum().item()) & 0xFFFFFFFFFFFFFFFF
    s2 = 0
    step = 1 << 24  # in pieces: the int64 temporaries of a 16384 x 2048 plane stay small
    for o in range(0, v.numel(), step):
        part = v[o:o + step]
        idx = torch.arange(o + 1, o + 1 + part.numel(), dtype=torch.int64, device=v.device)
        s2 = (s2 + int((part * idx).sum().item())) & 0xFFFFFFFFFFFFFFFF
    return [s1, s2]
