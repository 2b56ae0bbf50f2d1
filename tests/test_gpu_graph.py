"""The asynchronous device entries only enqueue kernels on the caller's stream once a context's buffers are sized, so a caller can capture an encode -- or a whole batch,
whose pipeline forks to a stream of the context's own and joins back through events -- into a HIP graph and replay it (the launch-bound case: small images, where the two
launches' gaps are 10-15 % of an encode).  Replays must reproduce the planes of the eager call, bit for bit, on fresh inputs of the same shape."""
import numpy as np
import pytest

from oracle.bind import PLANES

pytestmark = pytest.mark.gpu


def _host(planes):
    import torch
    return {k: v.cpu().numpy().view(np.uint32 if v.dtype == torch.int32 else np.uint8) for k, v in planes.items()}


def test_single_encode_in_a_hip_graph(oracle):
    import torch
    import limg_amd
    g = limg_amd.LimgHip(0)
    try:
        W, H = 512, 64
        a, b = oracle.photo_noise(W, H, 71), oracle.random_gradient(W, H, 72, True)
        img = torch.from_numpy(a.view(np.int32)).cuda()
        planes = g.alloc_planes_device(W, H)
        g.encode3d_device(img, True, planes)  # sizes the context's scratch and fills the noise table: nothing is allocated during capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g.encode3d_device(img, True, planes)
        for host in (a, b, a):
            img.copy_(torch.from_numpy(host.view(np.int32)))
            for v in planes.values():
                v.zero_()
            graph.replay()
            torch.cuda.synchronize()
            want = oracle.encode3d(host, True)
            got = _host(planes)
            bad = [k for k in PLANES if not np.array_equal(got[k], want[k])]
            assert not bad, bad
        g.check()
    finally:
        g.close()


def test_batch_pipeline_in_a_hip_graph(oracle):
    """17 images: the default rule runs them in sub-batches of 4 with the float stage of the next sub-batch on the context's second stream -- a fork / join inside the
    captured region."""
    import torch
    import limg_amd
    g = limg_amd.LimgHip(0)
    try:
        W, H, n = 256, 24, 17
        host = [oracle.photo_noise(W, H, 300 + i) for i in range(n)]
        imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
        outs = [g.alloc_planes_device(W, H) for _ in imgs]
        g.encode3d_batch_device(imgs, True, outs)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g.encode3d_batch_device(imgs, True, outs)
        for pl in outs:
            for v in pl.values():
                v.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for i, (h, pl) in enumerate(zip(host, outs)):
            want = oracle.encode3d(h, True)
            got = _host(pl)
            bad = [k for k in PLANES if not np.array_equal(got[k], want[k])]
            assert not bad, (i, bad)
        g.check()
    finally:
        g.close()
