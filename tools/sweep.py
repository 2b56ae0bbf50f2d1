#!/usr/bin/env python3
"""BASELINE.json configs[2]: 8192x8192 RGBA photo-noise on one MI355X -- adaptive sweep over errorFactor and forced-shift
sweep (bits per factor = 8 - shift on all three factors).  Prints a markdown table (Mpx/s from HIP-event kernel time, PSNR)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import limg_amd  # noqa: E402

import json  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json"))) if W == 8192 else {}  # the real reference's PSNR of every setting (tools/make_golden_fullsize.py)


def ref_psnr(name, psnr):
    e = GOLD.get(name)
    return "—" if e is None else ("%.4f %s" % (e["psnr"], "(equal)" if abs(e["psnr"] - psnr) < 1e-9 else "(DIFFERS)"))

g = limg_amd.LimgHip(0)
img = g.synth_device("photo_noise", W, W, seed=1)
planes = g.alloc_planes_device(W, W)


def run(ef=100, shift=None, steps=10):
    g.set_options(forced_shift=shift)
    for _ in range(2):
        g.encode3d_device(img, True, planes, error_factor=ef)
    torch.cuda.synchronize()
    g.profile_begin()
    for _ in range(steps):
        g.encode3d_device(img, True, planes, error_factor=ef)
    torch.cuda.synchronize()
    ms = float(g.profile_end(steps)[:, :2].sum(axis=1).mean())  # k_fit_tpb + k_encode_persistent
    psnr, _ = g.compare_device(img, planes["pDecoded"], True)
    sh = planes["pShiftABCX"].view(torch.int32)
    return ms, psnr


print("| setting | kernel ms | Mpx/s | 39 B/px / t (TB/s) | of the 8 TB/s roofline | perceptual PSNR (dB) | the reference's |")
print("|---|---|---|---|---|---|---|")
for ef in (0, 25, 50, 100, 200, 400):
    ms, psnr = run(ef=ef)
    print("| errorFactor %d | %.3f | %.0f | %.2f | %.3f | %.4f | %s |" % (ef, ms, W * W / ms / 1e3, 39 * W * W / ms / 1e9, 39 * W * W / ms / 8e12 * 1e3, psnr, ref_psnr("pn8192" if ef == 100 else "pn8192_ef%d" % ef, psnr)))
for bits in (8, 7, 6, 5, 4, 3, 2):
    s = 8 - bits
    ms, psnr = run(shift=(s, s, s))
    print("| forced %d bits/factor | %.3f | %.0f | %.2f | %.3f | %.4f | %s |" % (bits, ms, W * W / ms / 1e3, 39 * W * W / ms / 1e9, 39 * W * W / ms / 8e12 * 1e3, psnr, ref_psnr("pn8192_forced%d" % s, psnr)))
g.set_options()
g.check()
g.close()
