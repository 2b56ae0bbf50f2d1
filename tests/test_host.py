"""CPU-only checks of the product's host side: the C ABI exports every symbol include/limg_hip.h declares, the host-only
helpers (dither noise stream, chain walk, strip partition) agree with the oracle / golden vectors, and the library
refuses to work without a GPU instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import limg_amd
    if not os.path.exists(limg_amd.LIB_PATH):
        from limg_amd import build
        build.build_all()
    return limg_amd.load_library()


def test_abi_exports_every_declared_symbol(lib):
    import limg_amd
    hdr = open(os.path.join(ROOT, "include", "limg_hip.h")).read()
    inline = set(re.findall(r"static inline \w+ (limg_hip_[a-z0-9_]+)\s*\(", hdr))  # header-only wrappers (limg_hip_default_options): not exports
    declared = set(re.findall(r"\b(limg_hip_[a-z0-9_]+)\s*\(", hdr)) - inline
    assert declared, "no declarations parsed"
    assert declared == set(limg_amd.ABI_SYMBOLS), declared ^ set(limg_amd.ABI_SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None
    # the test-hooks header: declared there, exported by the test build (what this suite loads), absent from the product (tests/test_product_library.py)
    hooks = open(os.path.join(ROOT, "include", "limg_hip_test_hooks.h")).read()
    inline = set(re.findall(r"static inline \w+ (limg_hip_[a-z0-9_]+)\s*\(", hooks))
    declared = set(re.findall(r"\b(limg_hip_[a-z0-9_]+)\s*\(", hooks)) - inline
    assert declared == set(limg_amd.TEST_ABI_SYMBOLS), declared ^ set(limg_amd.TEST_ABI_SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None
    assert lib.limg_hip_version().decode().startswith("limg_hip")


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    assert lib.limg_hip_init(0, C.byref(ctx)) == 100  # limg_error_Generic: no device, no fallback
    assert not ctx


def test_product_does_not_reference_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "limg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle/" not in txt.replace("oracle/_ref", "").replace("nothing in oracle/", "") or f in ("__init__.py",), (f,)
                assert "import oracle" not in txt and "from oracle" not in txt, f
    # the two rsqrt tables (checker copy and product copy) are the same generated data
    a = open(os.path.join(ROOT, "oracle", "limg_rsqrt_x86_table.h")).read()
    b = open(os.path.join(ROOT, "limg_amd", "csrc", "limg_rsqrt_x86_table.h")).read()
    assert a == b


def test_noise_stream_matches_oracle_dither(lib, oracle):
    calls = 300
    tab = np.zeros((calls, 64), dtype=np.uint8)
    assert lib.limg_hip_host_noise_table(tab.ctypes.data_as(C.c_void_p), calls) == 0
    rng = np.random.default_rng(1)
    h = 0xCA7F00D15BADF00D
    for k in range(calls):
        f = rng.integers(0, 256, 64, dtype=np.uint8)
        s = int(rng.integers(1, 8))
        h2, want = oracle.dither(s, h, f)
        t = f.astype(np.int32) + ((tab[k].astype(np.int32) & ((1 << s) - 1)) - (1 << (s - 1)))
        got = (np.clip(t, 0, 255) >> s).astype(np.uint8)
        assert np.array_equal(got, want), k
        h = h2


@pytest.mark.parametrize("soft", [0, 1])
def test_chain_walk_known_answers(lib, soft):
    c = gu.chain()
    for n in (64, 16, 20, 40, 7, 15):
        h = 0xCA7F00D15BADF00D
        for want in c["aes_%d" % n]:
            buf = np.zeros(64, dtype=np.uint8)
            h = lib.limg_hip_host_chain_call(h, n, buf.ctypes.data_as(C.c_void_p), soft)
            assert "%016x" % h == want, (n, soft)


def test_chain_walk_partial_block_noise(lib, oracle):
    for n in (15, 16, 20, 7, 63, 40):
        rng = np.random.default_rng(n)
        f = rng.integers(0, 256, n, dtype=np.uint8)
        for s in range(1, 8):
            buf = np.zeros(64, dtype=np.uint8)
            h = lib.limg_hip_host_chain_call(0x1234567890ABCDEF, n, buf.ctypes.data_as(C.c_void_p), 0)
            oh, want = oracle.dither(s, 0x1234567890ABCDEF, f)
            t = f.astype(np.int32) + ((buf[:n].astype(np.int32) & ((1 << s) - 1)) - (1 << (s - 1)))
            assert h == oh and np.array_equal((np.clip(t, 0, 255) >> s).astype(np.uint8), want), (n, s)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_chain_walk_rectangles_of_any_size(lib, oracle, mode):
    """The merged-block encoder dithers whole rectangles: one call over N pixels = floor(N / 8) AES rounds + N % 8 PCG steps (or N PCG steps)."""
    for n in (65, 100, 1000, 4096 + 3, 64 * 37 + 5):
        rng = np.random.default_rng(n)
        f = rng.integers(0, 256, n, dtype=np.uint8)
        buf = np.zeros(n + 8, dtype=np.uint8)
        h = lib.limg_hip_host_chain_call(0xCA7F00D15BADF00D, n, buf.ctypes.data_as(C.c_void_p), mode)
        for s in (1, 4, 7):
            oh, want = oracle.dither(s, 0xCA7F00D15BADF00D, f, mode=(1 if mode == 2 else 0))
            t = f.astype(np.int32) + ((buf[:n].astype(np.int32) & ((1 << s) - 1)) - (1 << (s - 1)))
            assert h == oh and np.array_equal((np.clip(t, 0, 255) >> s).astype(np.uint8), want), (n, s, mode)


@pytest.mark.parametrize("size_y,pool", [(8192, 0), (8192, 2), (618, 8), (200, 2), (64, 3), (24, 8), (8, 1), (1000, 7)])
def test_partition_rule(lib, size_y, pool):
    cc, rr = C.c_uint32(), C.c_uint32()
    assert lib.limg_hip_host_partition(size_y, pool, C.byref(cc), C.byref(rr)) == 0
    # literal restatement of src/limg.cpp:2114-2134
    if pool == 0:
        want = (1, 0)
    else:
        tc = pool * 4
        yr = ((size_y // 8) // tc) * 8
        if yr == 0:
            tc = pool
            yr = ((size_y // 8) // tc) * 8
        want = (tc, yr // 8) if yr else (1, 0)
    assert (cc.value, rr.value) == want


def test_search_automaton_replays_reference_searches():
    """The generated decision table (tools/make_search_table.py) driven by the real reference's trial outcomes
    (tests/golden/blocks.npz: all 729 shift triples per block) ends at the real reference's search result."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_search_table as mst
    words = mst.encode(mst.build())
    hdr = open(os.path.join(ROOT, "limg_amd", "csrc", "limg_search_table.h")).read()
    assert "LIMG_SEARCH_STATES %d" % len(words) in hdr
    assert all(mst.ENTRY_FMT % w in hdr for w in words[:50] + words[-50:])
    z = gu.blocks()
    for bi in range(int(z["count"])):
        for ch in (4, 3):
            p = "b%02d_%d_" % (bi, ch)
            t = z[p + "trials"]
            shift, n = mst.walk(words, lambda a, b, c: bool(t[a, b, c, 0]))
            assert list(shift) == z[p + "search_100_1"].tolist(), p


def test_accurate_search_automaton_replays_reference_searches():
    """The accurate search's automaton (tools/make_search_table.py -> limg_search_table_accurate.h, expanded by the library for the kernel) driven by the real
    reference's trial outcomes and block errors (tests/golden/blocks.npz) ends at the real reference's accurate-search result; the committed header is what the
    generator produces."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_search_table as mst
    words = mst.encode_accurate(mst.build_accurate())
    hdr = open(os.path.join(ROOT, "limg_amd", "csrc", "limg_search_table_accurate.h")).read()
    assert "LIMG_SEARCH_ACC_STATES %d" % len(words) in hdr
    assert all(mst.ACC_ENTRY_FMT % w in hdr for w in words[:60] + words[len(words) // 2: len(words) // 2 + 60] + words[-60:])
    z = gu.blocks()
    longest = 0
    for bi in range(int(z["count"])):
        for ch in (4, 3):
            p = "b%02d_%d_" % (bi, ch)
            t = z[p + "trials"]
            shift, n = mst.walk_accurate(words, lambda a, b, c: (bool(t[a, b, c, 0]), int(t[a, b, c, 1])))
            assert list(shift) == z[p + "search_100_0"].tolist(), p
            longest = max(longest, n)
    assert longest > 40  # the walks are real searches, not early exits


def test_accurate_search_automaton_equals_loops_on_random_outcomes():
    """Random pass / fail outcomes and block errors: the table and a literal run of the restated loops (with the caller-side bookkeeping of the kernel) agree."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_search_table as mst
    words = mst.encode_accurate(mst.build_accurate())
    rnd = random.Random(7)
    for _ in range(600):
        pr = rnd.random()
        memo = {}

        def outcome(a, b, c):
            if (a, b, c) not in memo:
                memo[(a, b, c)] = (rnd.random() < pr, rnd.randrange(1, 5000))
            return memo[(a, b, c)]
        got, _n = mst.walk_accurate(words, outcome)
        g = mst.search_accurate()
        shift, min_be = (0, 0, 0), None
        try:
            t = next(g)
            while True:
                ok, be = outcome(*t[:3])
                if ok and (t[3] == 1 or be < min_be):
                    shift, min_be = t[:3], be
                t = g.send(ok)
        except StopIteration:
            pass
        assert got == shift


def test_search_automaton_equals_oracle_on_random_outcomes(oracle):
    """Random pass/fail oracles: the table and a literal re-run of the generator agree (exhaustive merge is sound)."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_search_table as mst
    words = mst.encode(mst.build())
    rnd = random.Random(5)
    for _ in range(2000):
        pr = rnd.random()
        memo = {}

        def outcome(a, b, c):
            if (a, b, c) not in memo:
                memo[(a, b, c)] = rnd.random() < pr
            return memo[(a, b, c)]
        got, _n = mst.walk(words, outcome)
        g = mst.search_fast()
        try:
            t = next(g)
            while True:
                t = g.send(outcome(*t))
        except StopIteration as e:
            want = e.value
        assert got == want


def test_pcg_chain_walk(lib, oracle):
    from oracle.bind import DITHER_PCG
    c = gu.chain()
    for n in (64, 16, 7):
        h = 0xCA7F00D15BADF00D
        for want in c["pcg_%d" % n]:
            buf = np.zeros(64, dtype=np.uint8)
            h = lib.limg_hip_host_chain_call(h, n, buf.ctypes.data_as(C.c_void_p), 2)
            assert "%016x" % h == want
    f = (np.arange(64) * 4 + 1).astype(np.uint8)
    for s in range(1, 8):
        buf = np.zeros(64, dtype=np.uint8)
        lib.limg_hip_host_chain_call(0xCA7F00D15BADF00D, 64, buf.ctypes.data_as(C.c_void_p), 2)
        t = f.astype(np.int32) + ((buf.astype(np.int32) & ((1 << s) - 1)) - (1 << (s - 1)))
        assert (np.clip(t, 0, 255) >> s).tolist() == c["dither_bytes"]["pcg_s%d" % s]


def test_host_blocked_match_predicate_equals_oracle(oracle):
    """The similarity predicate the host merge evaluates for far-apart pairs (liblimg_hip.so, no GPU) == the oracle's (pinned to the reference)."""
    import limg_amd
    lib = limg_amd.load_library()
    rng = np.random.default_rng(11)
    for channels, img in ((4, oracle.photo_noise(256, 128, 3)), (4, oracle.random_gradient(256, 128, 3, False)), (3, oracle.photo_noise(256, 128, 4))):
        recs = oracle.blocked_encode3d(img, channels == 4, planes=False)["pass1"].reshape(-1)
        seen = set()
        for _ in range(1500):
            i, j = rng.integers(0, recs.size, 2)
            if rng.random() < 0.7:
                j = min(recs.size - 1, i + int(rng.integers(1, 3)))
            a, b = recs[i:i + 1].copy(), recs[j:j + 1].copy()
            want = oracle.blocked_matches(channels, a, b)
            assert limg_amd.host_blocked_matches(channels, a, b, lib) == want, (channels, i, j)
            seen.add(want)
        assert seen == {True, False}


@pytest.mark.parametrize("use_bits", [True, False])
def test_host_blocked_merge_equals_oracle(oracle, use_bits):
    """The product's greedy merge (host, no GPU) over the oracle's pass-1 fits == the oracle's rectangles (pinned to the reference), both through the
    similarity-bit window and with on-demand evaluation."""
    import limg_amd
    lib = limg_amd.load_library()
    for alpha, img in ((True, oracle.photo_noise(512, 256, 3)), (True, oracle.random_gradient(384, 256, 3, True)), (False, oracle.photo_noise(203, 131, 4)),
                       (True, np.full((96, 160), 0xFF204060, dtype=np.uint32))):
        want = oracle.blocked_encode3d(img, alpha, planes=False)
        got = limg_amd.host_blocked_merge(want["pass1"], 4 if alpha else 3, use_bits, lib)
        assert len(got) == len(want["regions"])
        for f in ("ox", "oy", "rx", "ry"):
            assert np.array_equal(got[f], want["regions"][f]), f


def test_gather_offsets_and_chain_bases():
    """Host arithmetic of the multi-GPU entries (include/limg_hip.h "multi-GPU"): 16-byte aligned piece offsets of the variable-size stream gather and
    the exclusive prefix of the per-rank dither-call totals."""
    import limg_amd
    sizes = np.array([100, 0, 16, 4097, 1, 123456789012], dtype=np.uint64)
    offs = limg_amd.host_gather_offsets(sizes)
    assert offs.tolist() == [0, 112, 112, 128, 4240, 4256, 4256 + 123456789024]
    assert all(int(o) % 16 == 0 for o in offs)
    calls = np.array([5, 0, 7, 2 ** 40, 1], dtype=np.uint64)
    assert limg_amd.host_chain_bases(calls).tolist() == [0, 5, 5, 12, 12 + 2 ** 40]
    assert limg_amd.host_gather_offsets(np.array([33], dtype=np.uint64)).tolist() == [0, 48]


def test_noise_checkpoints_are_the_chain():
    """limg_amd/csrc/limg_noise_checkpoints.h (what the GPU fills the noise table from) against a fresh serial walk of the dither chain by the host implementation
    that tests/golden/chain.json pins to the reference: every one of the 16384 dense values (every 1024th call) and of the 2048 far values (every 65536th call, through
    2^27 calls), and the walk's known answers of SURVEY 8(c) on the way."""
    import ctypes as C
    import re
    import limg_amd
    L = limg_amd.load_library()
    text = open(os.path.join(ROOT, "limg_amd", "csrc", "limg_noise_checkpoints.h")).read()
    every = int(re.search(r"LIMG_NOISE_CHECKPOINT_EVERY (\d+)", text).group(1))
    count = int(re.search(r"LIMG_NOISE_CHECKPOINT_COUNT (\d+)", text).group(1))
    far_every = int(re.search(r"LIMG_NOISE_FAR_EVERY (\d+)", text).group(1))
    far_count = int(re.search(r"LIMG_NOISE_FAR_COUNT (\d+)", text).group(1))
    dense_text, far_text = text.split("#define LIMG_NOISE_FAR_INIT")
    vals = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", dense_text)], dtype=np.uint64)
    far = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", far_text)], dtype=np.uint64)
    assert every == 1024 and count == 16384 and vals.size == count
    assert far_every == 65536 and far_count == 2048 and far.size == far_count
    walked = np.zeros(far_count, dtype=np.uint64)
    L.limg_hip_host_chain_checkpoints(far_every * far_count, far_every, walked.ctypes.data_as(C.c_void_p), 0)  # 2^27 calls, serial: a few seconds
    assert np.array_equal(far, walked)
    walked = np.zeros(count, dtype=np.uint64)
    L.limg_hip_host_chain_checkpoints(every * count, every, walked.ctypes.data_as(C.c_void_p), 0)
    assert np.array_equal(vals, walked)
    assert np.array_equal(far[:every * count // far_every], vals[::far_every // every])
    first = np.zeros(5, dtype=np.uint64)
    L.limg_hip_host_chain_checkpoints(5, 1, first.ctypes.data_as(C.c_void_p), 0)
    assert [hex(int(v)) for v in first] == ["0xca7f00d15badf00d", "0x4ae914d5e23b0473", "0x1db0e1e7cd750f32", "0x13d534ac987485a9", "0xd82d4ba61c55878b"]  # SURVEY 8(c) chain KAT


def test_dense_checkpoints_beyond_the_embedded_table():
    """What a context uploads for an image of more than 5.59 M blocks (limg_hip_api.hip ensure_checkpoints): dense chain values beyond the embedded table's 16 Mi calls,
    made from the far table on host threads (limg_hip_host_dense_checkpoints) == the serial walk; across the boundary, over whole and partial far stretches; refused
    beyond the far table's reach."""
    import ctypes as C
    import limg_amd
    L = limg_amd.load_library()
    first, count = 16384 - 3, 64 * 3 + 10  # three dense values, then three far stretches and a bit
    got = np.zeros(count, dtype=np.uint64)
    assert L.limg_hip_host_dense_checkpoints(first, count, got.ctypes.data_as(C.c_void_p)) == 0
    walked = np.zeros(first + count, dtype=np.uint64)
    L.limg_hip_host_chain_checkpoints((first + count) * 1024, 1024, walked.ctypes.data_as(C.c_void_p), 0)
    assert np.array_equal(got, walked[first:])
    got = np.zeros(7, dtype=np.uint64)  # inside one far stretch, not at its start
    assert L.limg_hip_host_dense_checkpoints(16384 + 64 + 17, 7, got.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(got, walked[16384 + 64 + 17: 16384 + 64 + 24])
    assert L.limg_hip_host_dense_checkpoints(0, 0, got.ctypes.data_as(C.c_void_p)) == 0
    assert L.limg_hip_host_dense_checkpoints(2048 * 64 - 7, 7, got.ctypes.data_as(C.c_void_p)) == 0
    assert L.limg_hip_host_dense_checkpoints(2048 * 64 - 6, 7, got.ctypes.data_as(C.c_void_p)) == 103  # limg_hip_error_OutOfBounds
    assert L.limg_hip_host_dense_checkpoints(0, 1, None) == 102


def test_oracle_sanitizer_script_runs():
    """tools/oracle_sanitize.sh (the CPU restatements of both encoders under ASan + UBSan on the golden cases; GPU sanitizers are not available on the pool) still builds
    and passes -- it had rotted once (the merged-block restatement's file missing from its compile line)."""
    import shutil
    import subprocess
    if not shutil.which("gcc") or subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip() == "libasan.so":
        pytest.skip("no gcc / libasan here")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "oracle_sanitize.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "oracle clean under ASan+UBSan" in r.stdout
