import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, limg_amd
g = limg_amd.LimgHip(0)
img = g.synth_device("photo_noise", 8192, 8192, seed=1).cpu().numpy().view(np.uint32)
for name, fn in (("limg_hip_encode3d (11 planes D2H)", lambda: g.encode3d(img, True)), ("limg_hip_encode_stream (stream D2H)", lambda: g.encode_stream(img, True)),
                 ("limg_hip_encode3d_perf (H2D only)", lambda: g.encode3d_perf(img, True))):
    fn()
    t = time.perf_counter(); fn(); dt = time.perf_counter() - t
    print("%-40s %.1f ms  %.1f Mpx/s" % (name, dt * 1e3, 67.1 / dt))
