"""pack / decode kernel intervals of the stream path for the library LIMG_HIP_LIB names (same-box A/B of build variants)"""
import sys
sys.path.insert(0, ".")
import torch
import limg_amd
g = limg_amd.LimgHip(0)
for kind, W in (("photo_noise", 8192), ("random_gradient", 4096)):
    img = g.synth_device(kind, W, W, seed=1)
    st, n = g.encode_stream_device(img, True)
    for _ in range(3):
        g.encode_stream_device(img, True, out=st, want_size=False)
    torch.cuda.synchronize()
    g.profile_begin()
    for _ in range(20):
        g.encode_stream_device(img, True, out=st, want_size=False)
    torch.cuda.synchronize()
    k = g.profile_end(40)
    print(limg_amd.LIB_PATH, kind, W, "bytes", n, "pack ms %.4f (min %.4f)" % (float(k[1::2, 0].mean()), float(k[1::2, 0].min())), flush=True)
g.check()
