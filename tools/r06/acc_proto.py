#!/usr/bin/env python3
"""Driver of tools/r06/acc_proto.hip on the GPU box: the accurate search's trial + automaton alone, in the library's mapping (mode 0) and in the four-blocks-per-wave
early-exit mapping (mode 1), on the blocks of a W x H photo-noise image whose records / pre-dither factor bytes / expected shifts come from the CPU oracle.
usage: python tools/r06/acc_proto.py [--size 4096x1024] [--reps 20] [--wg-per-cu 6]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="4096x1024")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--wg-per-cu", type=int, default=6)
    ap.add_argument("--modes", default="0,1")
    args = ap.parse_args()
    W, H = (int(v) for v in args.size.split("x"))
    lib_path = os.path.join(HERE, "libacc_proto.so")
    if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < os.path.getmtime(os.path.join(HERE, "acc_proto.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", lib_path, os.path.join(HERE, "acc_proto.hip")])
    import torch
    from oracle.bind import Oracle
    L = C.CDLL(lib_path)
    L.acc_proto_run.restype = C.c_int
    L.acc_proto_run.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_int, C.c_void_p]
    orc = Oracle()
    img = orc.photo_noise(W, H, 1)
    want = orc.encode3d(img, True, fast=False, extras=True, planes=False, worker_threads=16)
    by, bx = H // 8, W // 8
    n = by * bx

    def blocks_of(a):  # (H, W) -> (n, 64), block-major, row-major inside a block
        return np.ascontiguousarray(a.reshape(by, 8, bx, 8).transpose(0, 2, 1, 3).reshape(n, 64))

    px = blocks_of(img)
    fac = blocks_of(want["preA"].astype(np.uint32) | (want["preB"].astype(np.uint32) << 8) | (want["preC"].astype(np.uint32) << 16))
    rec = np.zeros((n, 24), dtype=np.int16)
    r = want["records"].reshape(n)
    for i, k in enumerate(("dirA_min", "dirA_max", "dirB_offset", "dirB_mag", "dirC_offset", "dirC_mag")):
        rec[:, 4 * i:4 * i + 4] = r[k]
    sh = want["shifts"].reshape(n, 3).astype(np.uint32)
    expect = sh[:, 0] | (sh[:, 1] << 8) | (sh[:, 2] << 16)
    states = L.acc_proto_states()
    table = np.zeros(states * 8, dtype=np.uint32)
    L.acc_proto_table(table.ctypes.data_as(C.c_void_p))
    ef = 100
    max_pixel, max_block = 6 * (ef // 2) * 7, 4 * (ef // 2) * 7
    block_limit = (max_block * 64 + 15) // 16
    dev = lambda a, dt: torch.from_numpy(a.view(dt)).cuda()
    d_px, d_fac, d_rec, d_tab = dev(px, np.int32), dev(fac, np.int32), dev(rec, np.int16), dev(table, np.int32)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    wgs = cus * args.wg_per_cu
    print("image %dx%d photo-noise: %d blocks, %.2f trials per block (oracle), %d states, %d workgroups of 4 waves (%d per CU)" % (W, H, n, want["trials"] / n, states, wgs, args.wg_per_cu), flush=True)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for mode in (int(m) for m in args.modes.split(",")):
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        run = lambda: L.acc_proto_run(mode, C.c_void_p(d_px.data_ptr()), C.c_void_p(d_fac.data_ptr()), C.c_void_p(d_rec.data_ptr()), C.c_void_p(d_tab.data_ptr()),
                                      C.c_void_p(out.data_ptr()), n, max_pixel, block_limit, wgs, stream)
        assert run() == 0
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        bad = int((got != expect).sum())
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = min(ts)
        print("mode %d (%s): %d of %d blocks differ from the oracle's shifts; %.4f ms best, %.4f median = %.2f ns per block"
              % (mode, "lane == pixel, wave == block" if mode == 0 else "4 blocks per wave, 16 pixels per step, early exit", bad, n, min(ts), float(np.median(ts)), min(ts) * 1e6 / n), flush=True)
        assert bad == 0
    if 0 in res and 1 in res:
        print("mode 1 / mode 0 time: %.3f  (speed-up %.2f x)" % (res[1] / res[0], res[0] / res[1]))


if __name__ == "__main__":
    main()
