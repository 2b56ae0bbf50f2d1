"""limg_hip_encode3d (host pointers, pageable memory) at 8192^2: best of 5, for the library LIMG_HIP_LIB names"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, limg_amd
g = limg_amd.LimgHip(0)
W = 8192
img = g.synth_device("photo_noise", W, W, seed=1).cpu().numpy().view(np.uint32)
import ctypes as C
out = {k: np.zeros((W, W), dtype=np.uint32) for k in limg_amd.P32}
out.update({k: np.zeros((W, W), dtype=np.uint8) for k in limg_amd.P8})
info = limg_amd.Info(*[out[k].ctypes.data for k in limg_amd.PLANES])
for pool in (0, 2):
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        r = g.lib.limg_hip_encode3d(g.ctx, img.ctypes.data_as(C.c_void_p), W, W, 1, C.byref(info), 100, pool, 1)
        ts.append((time.perf_counter() - t) * 1e3)
        assert r == 0
    print(limg_amd.LIB_PATH, "pool", pool, "ms", " ".join("%.2f" % t for t in ts), flush=True)
ts = []
for _ in range(6):
    t = time.perf_counter()
    r = g.lib.limg_hip_encode3d_perf(g.ctx, img.ctypes.data_as(C.c_void_p), W, W, 1, 100, 0, 1)
    ts.append((time.perf_counter() - t) * 1e3)
    assert r == 0
print(limg_amd.LIB_PATH, "limg_hip_encode3d_perf (host pointer, nothing stored) ms", " ".join("%.2f" % t for t in ts), flush=True)
