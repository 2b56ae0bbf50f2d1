"""PCIe-inclusive rates of the host-pointer entry points (what the shim's limg_encode3d_test / limg_encode / limg_encode3d_test_perf call) on 8192^2 photo-noise.
Two figures per entry: with caller buffers whose pages already exist (the library's own cost: H2D, kernels, D2H), and -- for the plane entry -- with freshly
allocated, never-touched caller planes (what a caller that mallocs and calls pays on top: the kernel faulting in 2.35 GB of zero pages during the copy;
the reference's CPU path pays the same first touch).  Prints JSON lines.  Never the headline `value` of bench.py."""
import ctypes as C
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
import limg_amd

W = H = 8192
g = limg_amd.LimgHip(0)
img = g.synth_device("photo_noise", W, H, seed=1).cpu().numpy().view(np.uint32)


def planes(touch):
    out = {k: np.empty((H, W), dtype=np.uint32 if k in limg_amd.P32 else np.uint8) for k in limg_amd.PLANES}
    if touch:
        for v in out.values():
            v.fill(0)
    return out


def encode_into(out):
    info = limg_amd.Info(*[out[k].ctypes.data for k in limg_amd.PLANES])
    r = g.lib.limg_hip_encode3d(g.ctx, img.ctypes.data_as(C.c_void_p), W, H, 1, C.byref(info), 100, 0, 1)
    assert r == 0, r


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts)


warm = planes(True)
encode_into(warm)
t_touched = best(lambda: encode_into(warm))
fresh = [planes(False) for _ in range(2)]
t_fresh = min(best(lambda p=p: encode_into(p), 1) for p in fresh)
del fresh
t_stream = best(lambda: g.encode_stream(img, True))
t_perf = best(lambda: g.encode3d_perf(img, True))
px = W * H
for name, t, note in (("limg_hip_encode3d, caller planes already touched", t_touched, "268 MB in, 2.35 GB of planes out"),
                      ("limg_hip_encode3d, freshly allocated caller planes", t_fresh, "same + first touch of 2.35 GB of caller pages"),
                      ("limg_hip_encode_stream", t_stream, "268 MB in, ~123 MB stream out (incl. the wrapper's worst-case buffer)"),
                      ("limg_hip_encode3d_perf", t_perf, "H2D only")):
    print(json.dumps({"entry": name, "ms": round(t * 1e3, 2), "Mpixels_per_s": round(px / t / 1e6, 1), "note": note}), flush=True)
