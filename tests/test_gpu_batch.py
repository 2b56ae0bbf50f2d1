"""limg_hip_encode3d_batch_device: a list of images of one shape through ONE launch pair (the reference's per-file loop, src/main.cpp:278-323; BASELINE configs 2
and 4).  Every image must get exactly the planes of a single encode -- checked against the CPU oracle at small sizes and against eight single encodes at 4096^2."""
import numpy as np
import pytest

from oracle.bind import PLANES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import limg_amd
    g = limg_amd.LimgHip(0)
    yield g
    g.check()
    g.close()


def _host(planes):
    import torch
    return {k: v.cpu().numpy().view(np.uint32 if v.dtype == torch.int32 else np.uint8) for k, v in planes.items()}


@pytest.mark.parametrize("alpha,pool,fast,ef", [(True, 0, True, 100), (False, 0, True, 100), (True, 2, True, 25), (True, 0, False, 100), (True, 1, True, 0)])
def test_batch_equals_oracle(gpu, oracle, alpha, pool, fast, ef):
    """5 different images of 512 x 72 (2 work strips x 9 block rows each; a strip partition that restarts the chain inside every image): every plane of every
    image against the oracle's single-image encode."""
    import torch
    W, H = 512, 72
    host = [oracle.photo_noise(W, H, 40 + i) if i % 2 == 0 else oracle.random_gradient(W, H, 40 + i, i != 3) for i in range(5)]
    imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
    outs = [gpu.alloc_planes_device(W, H) for _ in imgs]
    gpu.encode3d_batch_device(imgs, alpha, outs, error_factor=ef, pool_threads=pool, fast=fast)
    torch.cuda.synchronize()
    gpu.check()
    for i, (h, pl) in enumerate(zip(host, outs)):
        want = oracle.encode3d(h, alpha, error_factor=ef, pool_threads=pool, fast=fast)
        got = _host(pl)
        bad = [(k, int((got[k] != want[k]).sum())) for k in PLANES if not np.array_equal(got[k], want[k])]
        assert not bad, (i, bad)


def test_batch_chunks_and_fallbacks(gpu, oracle):
    """A list longer than one launch pair takes (test hook: 3 images per pair => 3 + 3 + 1), a list of one, and the shapes that fall back to one encode per
    image (partial edge blocks) give the same planes."""
    import torch
    for (W, H, n, chunk) in ((256, 16, 7, 3), (256, 16, 1, 0), (61, 27, 3, 0)):
        host = [oracle.photo_noise(W, H, 90 + i) for i in range(n)]
        imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
        outs = [gpu.alloc_planes_device(W, H) for _ in imgs]
        gpu.set_options(test_batch_chunk=chunk)
        try:
            gpu.encode3d_batch_device(imgs, True, outs)
        finally:
            gpu.set_options()
        torch.cuda.synchronize()
        gpu.check()
        for i, (h, pl) in enumerate(zip(host, outs)):
            want = oracle.encode3d(h, True)
            got = _host(pl)
            bad = [(k, int((got[k] != want[k]).sum())) for k in PLANES if not np.array_equal(got[k], want[k])]
            assert not bad, (W, H, i, bad)


@pytest.mark.parametrize("sub", [1, 2, 3])
@pytest.mark.parametrize("alpha,pool,fast,ef", [(True, 0, True, 100), (False, 2, True, 25), (True, 0, False, 100)])
def test_batch_as_pipeline_of_sub_batches(gpu, oracle, sub, alpha, pool, fast, ef):
    """limg_hip_options.batch_sub_images: the list as a pipeline of launch pairs -- the float stage of sub-batch k + 1 on the context's own stream next to the
    persistent kernel of sub-batch k (5 workgroups per CU) -- must give every image the planes of its single encode: 7 images in sub-batches of 1 / 2 / 3
    (the last one short), twice in a row (the second pass reuses tickets, descriptors and events)."""
    import torch
    W, H = 512, 72
    host = [oracle.photo_noise(W, H, 60 + i) if i % 2 == 0 else oracle.random_gradient(W, H, 60 + i, i != 3) for i in range(7)]
    imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
    outs = [gpu.alloc_planes_device(W, H) for _ in imgs]
    want = [oracle.encode3d(h, alpha, error_factor=ef, pool_threads=pool, fast=fast) for h in host]
    gpu.set_options(batch_sub_images=sub)
    try:
        for rep in range(2):
            for pl in outs:
                for v in pl.values():
                    v.zero_()
            gpu.encode3d_batch_device(imgs, alpha, outs, error_factor=ef, pool_threads=pool, fast=fast)
            torch.cuda.synchronize()
            gpu.check()
            for i, pl in enumerate(outs):
                got = _host(pl)
                bad = [(k, int((got[k] != want[i][k]).sum())) for k in PLANES if not np.array_equal(got[k], want[i][k])]
                assert not bad, (sub, rep, i, bad)
    finally:
        gpu.set_options()


@pytest.mark.parametrize("n", [17, 35])
def test_batch_default_rule_pipelines_long_lists(gpu, oracle, n):
    """Default options: a list of 16 ... 31 images goes in sub-batches of 4, of 32 and more in sub-batches of 8 (limg_hip_options.batch_sub_images = 0); every image
    still gets the planes of its single encode."""
    import torch
    W, H = 256, 24
    host = [oracle.photo_noise(W, H, 200 + i) if i % 3 else oracle.random_gradient(W, H, 200 + i, True) for i in range(n)]
    imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
    outs = [gpu.alloc_planes_device(W, H) for _ in imgs]
    gpu.set_options()
    gpu.encode3d_batch_device(imgs, True, outs)
    torch.cuda.synchronize()
    gpu.check()
    for i, (h, pl) in enumerate(zip(host, outs)):
        want = oracle.encode3d(h, True)
        got = _host(pl)
        bad = [(k, int((got[k] != want[k]).sum())) for k in PLANES if not np.array_equal(got[k], want[k])]
        assert not bad, (n, i, bad)


def test_batch_stats_cover_the_whole_list(gpu, oracle):
    """limg_hip_last_stats after a batched encode: all images of the list together, also when the list took several launch pairs (chunks) or a pipeline of
    sub-batches (only the last chunk was counted)."""
    import torch
    W, H = 256, 16
    host = [oracle.photo_noise(W, H, 70 + i) for i in range(7)]
    imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
    outs = [gpu.alloc_planes_device(W, H) for _ in imgs]
    want = np.zeros(30, dtype=np.uint64)
    for h in host:
        sh = oracle.encode3d(h, True, extras=True)["shifts"]
        for f in range(3):
            for s in range(9):
                n = int((np.minimum(sh[:, :, f], 8) == s).sum()) * 64
                want[3 + 9 * f + s] += n
                want[f] += (8 - s) * n
    for kw in (dict(), dict(batch_sub_images=-1), dict(test_batch_chunk=3), dict(batch_sub_images=2), dict(test_batch_chunk=4, batch_sub_images=3)):
        gpu.set_options(collect_stats=True, **kw)
        try:
            gpu.encode3d_batch_device(imgs, True, outs)
            cnt, px = gpu.last_stats()
        finally:
            gpu.set_options()
        assert px == 7 * W * H and np.array_equal(cnt, want), (kw, px, cnt, want)


def test_batch_argument_checks(gpu):
    import ctypes as C
    import torch
    import limg_amd
    img = torch.zeros((16, 256), dtype=torch.int32, device="cuda")
    pl = gpu.alloc_planes_device(256, 16)
    ins = (C.c_void_p * 2)(img.data_ptr(), None)
    infos = (limg_amd.Info * 2)(*[limg_amd.Info(*[pl[k].data_ptr() for k in limg_amd.PLANES]) for _ in range(2)])
    L = gpu.lib
    assert L.limg_hip_encode3d_batch_device(gpu.ctx, 2, ins, 256, 16, 1, infos, 100, 0, 1, None) == 102  # ArgumentNull: image 1 has no input
    assert L.limg_hip_encode3d_batch_device(gpu.ctx, 2, None, 256, 16, 1, infos, 100, 0, 1, None) == 102
    assert L.limg_hip_encode3d_batch_device(gpu.ctx, 0, ins, 256, 16, 1, infos, 100, 0, 1, None) == 0    # an empty list is no work
    assert L.limg_hip_encode3d_batch_device(gpu.ctx, 1, ins, 0, 16, 1, infos, 100, 0, 1, None) == 101   # InvalidParameter
    torch.cuda.synchronize()


def test_batch_of_8_at_4096_equals_single_encodes(gpu, oracle):
    """BASELINE config 4 as one GPU sees it (8 x 4096^2 random-gradient images, seeds 1..8): the batch call against eight single encodes of the same context,
    every plane compared on the device; image 0's first band against the oracle and its PSNR against the reference's figure."""
    import torch
    W = 4096
    imgs = [gpu.synth_device("random_gradient", W, W, seed=1 + i) for i in range(8)]
    outs = [gpu.alloc_planes_device(W, W) for _ in range(8)]
    gpu.encode3d_batch_device(imgs, True, outs)
    torch.cuda.synchronize()
    gpu.check()
    single = gpu.alloc_planes_device(W, W)
    for i in range(8):
        gpu.encode3d_device(imgs[i], True, single)
        torch.cuda.synchronize()
        for k in PLANES:
            assert torch.equal(single[k], outs[i][k]), (i, k)
    band = imgs[0][:128].cpu().numpy().view(np.uint32)
    want = oracle.encode3d(band, True)
    for k in PLANES:
        got = outs[0][k][:128].cpu().numpy()
        got = got.view(np.uint32) if got.dtype == np.int32 else got
        assert np.array_equal(got, want[k]), k
    psnr, _ = gpu.compare_device(imgs[0], outs[0]["pDecoded"], True)
    assert abs(psnr - 50.38) < 0.05
    # and again with photo-noise content (long searches), 4 images, repeated: the second pass must not see anything of the first
    imgs = [gpu.synth_device("photo_noise", W, W, seed=11 + i) for i in range(4)]
    for rep in range(2):
        gpu.encode3d_batch_device(imgs, True, outs[:4])
        torch.cuda.synchronize()
        for i in range(4):
            gpu.encode3d_device(imgs[i], True, single)
            torch.cuda.synchronize()
            for k in PLANES:
                assert torch.equal(single[k], outs[i][k]), (rep, i, k)
    gpu.check()
    # the same list as a pipeline of sub-batches of 2 and of 3 (float stage of the next sub-batch next to the persistent kernel of the current one)
    for sub in (2, 3):
        for pl in outs[:4]:
            for v in pl.values():
                v.zero_()
        gpu.set_options(batch_sub_images=sub)
        try:
            gpu.encode3d_batch_device(imgs, True, outs[:4])
        finally:
            gpu.set_options()
        torch.cuda.synchronize()
        for i in range(4):
            gpu.encode3d_device(imgs[i], True, single)
            torch.cuda.synchronize()
            for k in PLANES:
                assert torch.equal(single[k], outs[i][k]), ("sub", sub, i, k)
    gpu.check()
    del imgs, outs, single
    torch.cuda.empty_cache()
