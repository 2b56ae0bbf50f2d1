"""Several persistent kernels on one GPU at once, and a persistent kernel beside a foreign one: the look-back's progress guarantee.

The reference is re-entrant at any size (stack scratch only, src/limg.cpp:1890-1893; strips on a thread pool always complete, :2131-2136).  Here a context owns
device scratch, and include/limg_hip.h promises "one context per HIP stream / thread ... contexts are independent".  Round 4 broke that promise without a test
noticing: workgroup i of the persistent kernel took strip i without drawing a ticket, so with two such kernels in flight a resident workgroup
waited in its look-back for strips whose workgroups were never dispatched -- both kernels spun into the look-back's bound (seconds) and then dithered from a wrong
chain base.  The only two-context test used 512 x 256 images (fewer strips than residency slots) and could not see it.  These tests run at the BASELINE sizes,
compare EVERY plane of EVERY encode with a single-context encode of the same image on the device, require a clean status word and bound the wall time.
"""
import ctypes as C
import os
import time

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _planes_equal_count(torch, got, want):
    """number of planes that differ, as a device scalar (no host synchronisation: the comparison rides the stream behind the encode)"""
    bad = None
    for k in want:
        ne = (got[k] != want[k]).any().to(torch.int32)
        bad = ne if bad is None else bad + ne
    return bad


@pytest.fixture(scope="module")
def ctxs():
    import limg_amd
    gs = [limg_amd.LimgHip(0) for _ in range(4)]
    yield gs
    for g in gs:
        g.close()


@pytest.mark.parametrize("kind,W,H", [("photo_noise", 8192, 8192), ("random_gradient", 4096, 4096)])
def test_two_contexts_two_streams_full_size(ctxs, kind, W, H):
    """(a) two contexts x two HIP streams, 12 encodes alternating, both persistent kernels in flight together: every plane of every encode == the
    single-context result, limg_hip_check_device_status clean on both, wall time bounded (a look-back that runs into its bound costs seconds)."""
    import torch
    ref_ctx, g0, g1 = ctxs[0], ctxs[1], ctxs[2]
    img = ref_ctx.synth_device(kind, W, H, seed=1)
    want = ref_ctx.alloc_planes_device(W, H)
    ref_ctx.encode3d_device(img, True, want)
    torch.cuda.synchronize()
    ref_ctx.check()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [g0.alloc_planes_device(W, H), g1.alloc_planes_device(W, H)]
    bad = [None, None]
    # warm both contexts (scratch allocation, noise table) outside the timed, concurrent part
    for g, s, o in zip((g0, g1), streams, outs):
        with torch.cuda.stream(s):
            g.encode3d_device(img, True, o)
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(6):
        for i, (g, s, o) in enumerate(zip((g0, g1), streams, outs)):
            with torch.cuda.stream(s):
                for k in o:
                    o[k].fill_(0x5A if o[k].dtype == torch.uint8 else 0x5A5A5A5A)  # a stale plane cannot pass for a fresh one
                g.encode3d_device(img, True, o)
                b = _planes_equal_count(torch, o, want)
                bad[i] = b if bad[i] is None else bad[i] + b
    torch.cuda.synchronize()
    wall = time.time() - t0
    g0.check()
    g1.check()
    assert int(bad[0].item()) == 0 and int(bad[1].item()) == 0, (kind, int(bad[0].item()), int(bad[1].item()))
    assert wall < 10.0, wall
    del outs, want, img
    torch.cuda.empty_cache()


def test_three_contexts_batched_lists(ctxs):
    """(b) limg_hip_encode3d_batch_device lists on three contexts at once (bench.py --contexts 3, config 4): lists of 16 x 4096^2 gradient images go through the
    sub-batch pipeline (k_fit_tpb of sub-batch k + 1 beside the persistent kernel of sub-batch k) on three streams together; every image == its single encode."""
    import torch
    W, N = 4096, 16
    ref_ctx = ctxs[0]
    imgs = [ref_ctx.synth_device("random_gradient", W, W, seed=1 + i) for i in range(N)]
    want = [ref_ctx.alloc_planes_device(W, W) for _ in range(N)]
    for i in range(N):
        ref_ctx.encode3d_device(imgs[i], True, want[i])
    torch.cuda.synchronize()
    ref_ctx.check()
    gs = ctxs[1:4]
    streams = [torch.cuda.Stream() for _ in gs]
    outs = [[g.alloc_planes_device(W, W) for _ in range(N)] for g in gs]
    for g, s, o in zip(gs, streams, outs):  # warm
        with torch.cuda.stream(s):
            g.encode3d_batch_device(imgs, True, o)
    torch.cuda.synchronize()
    bad = [None] * len(gs)
    t0 = time.time()
    for it in range(2):
        for i, (g, s, o) in enumerate(zip(gs, streams, outs)):
            with torch.cuda.stream(s):
                for pl in o:
                    for k in pl:
                        pl[k].zero_()
                g.encode3d_batch_device(imgs, True, o)
                for j in range(N):
                    b = _planes_equal_count(torch, o[j], want[j])
                    bad[i] = b if bad[i] is None else bad[i] + b
    torch.cuda.synchronize()
    wall = time.time() - t0
    for g in gs:
        g.check()
    assert [int(b.item()) for b in bad] == [0] * len(gs)
    assert wall < 20.0, wall
    del outs, want, imgs
    torch.cuda.empty_cache()


def test_encode_beside_foreign_kernel(ctxs):
    """(c) an encode while a foreign kernel owns half the CUs (128 workgroups x 144 KiB of LDS, 8 ms) on another stream: part of the persistent grid cannot become
    resident until the foreign kernel ends; the resident workgroups draw every strip from the ticket and finish the image -- same planes, clean status."""
    import torch
    lib = C.CDLL(os.path.join(HERE, "helpers", "liboccupy.so"))  # built by __graft_entry__.build(); raises if missing
    lib.occupy_launch.restype = C.c_int
    lib.occupy_launch.argtypes = [C.c_int, C.c_int, C.c_void_p]
    W = 8192
    g = ctxs[1]
    img = g.synth_device("photo_noise", W, W, seed=1)
    want = g.alloc_planes_device(W, W)
    got = g.alloc_planes_device(W, W)
    g.encode3d_device(img, True, want)
    torch.cuda.synchronize()
    g.check()
    fs, es = torch.cuda.Stream(), torch.cuda.Stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for n_wg, us in ((128, 8000), (256, 5000), (64, 20000)):
        assert lib.occupy_launch(n_wg, us, C.c_void_p(fs.cuda_stream)) == 0
        with torch.cuda.stream(es):
            ev0.record()
            g.encode3d_device(img, True, got)
            ev1.record()
        torch.cuda.synchronize()
        g.check()
        for k in want:
            assert torch.equal(got[k], want[k]), (n_wg, us, k)
        assert ev0.elapsed_time(ev1) < 1000.0, (n_wg, us, ev0.elapsed_time(ev1))  # milliseconds: it may wait for the foreign kernel, never for a spin bound
    del got, want, img
    torch.cuda.empty_cache()


def test_lookback_timeout_is_loud(ctxs):
    """A protocol failure must not produce wrong planes silently.  Test hook: one strip never publishes its dither-call count, with a short spin bound.  Then:
    limg_hip_check_device_status reports it once; every strip of the chain behind the silent one leaves its chain-dependent planes (pDecoded, factors) untouched
    -- the sentinel stays -- while the strips before it are complete and correct; the next encode of the context is clean again."""
    import torch
    import limg_amd
    W, H = 2048, 512          # 8 strips per block row, 64 block rows: 512 work strips
    skip = 100
    g = ctxs[3]
    img = g.synth_device("photo_noise", W, H, seed=5)
    want = g.alloc_planes_device(W, H)
    g.encode3d_device(img, True, want)
    torch.cuda.synchronize()
    g.check()
    got = g.alloc_planes_device(W, H)
    for k in got:
        got[k].fill_(0x5A if got[k].dtype == torch.uint8 else 0x5A5A5A5A)
    g.set_options(test_lookback_spins=2000, test_skip_publish_strip=skip + 1)
    try:
        t0 = time.time()
        g.encode3d_device(img, True, got)
        torch.cuda.synchronize()
        assert time.time() - t0 < 30.0
        with pytest.raises(limg_amd.LimgHipError):
            g.check()
        g.check()  # reported once
    finally:
        g.set_options()
    row, col = (skip // 8) * 8, (skip % 8) * 256  # the silent strip's pixels: rows row..row+7, columns col..col+255
    for k in ("pDecoded", "pFactorsA", "pFactorsB", "pFactorsC"):
        # everything up to and including the silent strip is complete and right (the silent strip itself knows its own base)
        assert torch.equal(got[k][:row], want[k][:row]), k
        assert torch.equal(got[k][row:row + 8, :col + 256], want[k][row:row + 8, :col + 256]), k
        # everything behind it in the chain is untouched
        sent = 0x5A if got[k].dtype == torch.uint8 else 0x5A5A5A5A
        assert bool((got[k][row + 8:] == sent).all()), k
        assert bool((got[k][row:row + 8, col + 256:] == sent).all()), k
    for k in ("pShiftABCX", "pColAMin", "pColCMax"):  # the block-uniform planes: complete up to the silent strip, rows 0..3 of every block behind it (stored before the look-back)
        assert torch.equal(got[k][:row], want[k][:row]), k
        assert torch.equal(got[k].view(H // 8, 8, W)[:, :4], want[k].view(H // 8, 8, W)[:, :4]), k
    g.encode3d_device(img, True, got)
    torch.cuda.synchronize()
    g.check()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_width_ragged_encodes_on_three_threads(oracle):
    """The width-ragged path blocks its calling thread (the host walks the dither chain while the GPU works around it in bands): a service runs it from several threads, each
    with its own context and stream.  Three threads x 4 encodes of a 2046 x 1024 image (banded pipeline: 256 x 128 blocks) and, on one of them, the pool-of-2 variant
    (parallel chain walks): every result equals the oracle's."""
    import threading
    import numpy as np
    import torch
    import limg_amd
    from oracle.bind import PLANES
    W, H = 2046, 1024
    img = oracle.photo_noise(W, H, 61)
    want = {0: oracle.encode3d(img, True, worker_threads=8), 2: oracle.encode3d(img, True, pool_threads=2, worker_threads=8)}
    d_img = torch.from_numpy(img.view(np.int32)).cuda()
    errs = []

    def work(k):
        try:
            torch.cuda.set_device(0)
            g = limg_amd.LimgHip(0)
            st = torch.cuda.Stream()
            planes = g.alloc_planes_device(W, H)
            with torch.cuda.stream(st):
                for it in range(4):
                    pool = 2 if (k == 1 and it % 2 == 1) else 0
                    g.encode3d_device(d_img, True, planes, pool_threads=pool)
                    st.synchronize()
                    bad = [p for p in PLANES if not np.array_equal(planes[p].cpu().numpy().view(np.uint32 if planes[p].dtype == torch.int32 else np.uint8), want[pool][p])]
                    if bad:
                        errs.append((k, it, pool, bad))
            g.check()
            g.close()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
