"""bench.py's bookkeeping that the judge reads: counter entries of another round's build are refused (lines that quoted round-2 counters for round-3
kernels), `roofline.instruction_floor` is the stated formula, and `--size WxH` parses.  CPU only: nothing here touches a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pmc_entries_of_another_round_are_refused(tmp_path, monkeypatch):
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps({"fresh": {"source": "prof_%s_final" % bench.PMC_ROUND, "valu_instr_per_launch": 1.0e9, "fetch_kib": 1.0, "write_kib": 2.0},
                             "stale": {"source": "prof_r02_final4_fast", "valu_instr_per_launch": 2.0e9},
                             "nosource": {"valu_instr_per_launch": 3.0e9}}))
    monkeypatch.setattr(bench, "PMC_FILE", str(f))
    assert bench.pmc_entry("fresh")["valu_instr_per_launch"] == 1.0e9 and bench.pmc_stale_source("fresh") is None
    assert bench.pmc_entry("stale") is None and bench.pmc_stale_source("stale") == "prof_r02_final4_fast"
    assert bench.pmc_entry("nosource") is None
    assert bench.pmc_entry("absent") is None and bench.pmc_stale_source("absent") is None


def test_committed_pmc_file_is_this_rounds():
    d = json.load(open(bench.PMC_FILE))
    assert d, "profiles/pmc_by_workload.json is empty"
    for k, e in d.items():
        assert ("_%s" % bench.PMC_ROUND) in e["source"], (k, e["source"])
    assert "8192x8192_photo_noise_ef100_fused" in d  # the default line's key


def test_instruction_floor_formula():
    # 713.4 instructions per block x 1 Mi blocks at 578 G/s = 1.294 ms; 2 617 245 696 B / 1.294 ms = 2.022 TB/s = 0.2528 of 8 TB/s (the round-4 default line)
    pmc = {"valu_instr_per_launch": 713.4 * 1048576}
    assert abs(bench.instruction_floor(pmc, 39 * 8192 * 8192) - 0.2528) < 5e-4
    assert bench.instruction_floor(None, 1) is None and bench.instruction_floor({}, 1) is None


def test_size_argument_forms():
    assert bench.parse_size("8192") == (8192, 8192)
    assert bench.parse_size(4096) == (4096, 4096)
    assert bench.parse_size("8192x8190") == (8192, 8190)
    assert bench.parse_size("1024x618") == (1024, 618)


def test_failed_leg_is_on_the_line_and_fails_the_process(capsys, monkeypatch):
    """a look-back timeout sat in config.two_streams.error of three committed lines while bench.py exited 0.  Now every failed leg is in `errors`,
    and a non-empty list turns the exit status non-zero (bench.py's __main__)."""
    monkeypatch.setattr(bench, "ERRORS", [])
    monkeypatch.setattr(bench, "provenance", lambda: {"head": "x", "lib_sha": "y", "src_sha": "z", "bench_sha": "w"})
    r = bench.leg_failed("two_streams", RuntimeError("boom"))
    assert "boom" in r["error"]
    bench.emit({"metric": "m", "config": {"two_streams": r}})
    out = capsys.readouterr()
    line = json.loads(out.out.strip())
    assert line["errors"] == [{"leg": "two_streams", "error": "RuntimeError('boom')"}] and line["lib_sha"] == "y"
    assert "FAILED" in out.err
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "if ERRORS:\n        sys.exit(3)" in src


def test_source_sha_is_stable_and_sees_the_kernels():
    a = bench.source_sha()
    assert a == bench.source_sha() and len(a) == 64


def _final_lines():
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_bench_final*.json" % bench.PMC_ROUND))):
        for raw in open(f):
            if raw.strip().startswith("{"):
                out.append((os.path.basename(f), json.loads(raw)))
    return out


def test_final_bench_lines_are_one_build_without_failed_legs():
    """Evidence that cannot go stale silently: every `profiles/<round>_bench_final*.json` line carries the build it was measured on; all of them were measured on ONE
    build (same lib_sha, same src_sha) and none has a failed leg or an "error" anywhere."""
    lines = _final_lines()
    assert lines, "no profiles/%s_bench_final*.json yet" % bench.PMC_ROUND
    shas = {(l.get("lib_sha"), l.get("src_sha")) for _, l in lines}
    assert len(shas) == 1 and None not in next(iter(shas)), shas
    for name, l in lines:
        assert l.get("errors") == [], (name, l.get("errors"))
        assert '"error"' not in json.dumps(l), name
        assert l.get("head"), name


def test_profiles_index_is_current():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profiles_index.py"), "--check"])
    assert r.returncode == 0, "profiles/README.md's index of this round's bench lines is out of date: run python tools/profiles_index.py"


def test_isa_budget_tool_runs():
    """tools/isa_budget.py attributes the compiler's assembly to phases through markers in the kernel source; a reworded comment broke it silently in round 4.
    It must run on the committed sources and find both kernels (hipcc cross-compiles without a GPU)."""
    import subprocess
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import pytest
        pytest.skip("needs hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_budget.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-500:]
    assert "k_encode_persistent<4, false, true, false>" in r.stdout and "k_fit_tpb<4, false, true>" in r.stdout and "**sum**" in r.stdout


def test_sum64_checksum_matches_the_golden_generator():
    """bench.py --verify-golden compares a position-sensitive 64-bit checksum computed with torch on the device against the one tools/make_golden_fullsize.py computed
    with numpy from the real reference's planes: the two implementations must agree (uint32 words with the top bit set, bytes, lengths that cross the piece size)."""
    import importlib.util
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(ROOT, "tools", "make_golden_fullsize.py"))
    src = open(spec.origin).read()
    ns = {}
    exec(src[src.index("def sum64(a):"):src.index("def make_input(")], ns)  # (the module's import of oracle._ref is not wanted here)
    rng = np.random.default_rng(5)
    for n, dt in ((1, np.uint32), (1000, np.uint32), ((1 << 24) + 77, np.uint32), (4097, np.uint8), ((1 << 24) + 5, np.uint8)):
        a = rng.integers(0, 256 if dt == np.uint8 else (1 << 32), n, dtype=np.uint64).astype(dt)
        t = torch.from_numpy(a.view(np.int32) if dt == np.uint32 else a)
        assert bench.sum64_device(t) == ns["sum64"](a), (n, dt)


class _FakeDist:
    """torch.distributed as one rank of a world of one sees it (backend "nccl"): enough for bench.collective_evidence"""

    def __init__(self, world=1):
        self.world = world

    def get_backend(self):
        return "nccl"

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return 0

    def all_gather(self, out, t):
        for o in out:
            o.copy_(t)


class _FakeCtx:
    def __init__(self, mode):
        self.mode = mode

    def comm_init_from_torch(self, dist):
        import time
        if self.mode == "raise":
            raise RuntimeError("ncclCommInitRank failed")
        if self.mode == "hang":
            time.sleep(30)

    def comm_info(self):
        return {"ranks": 1, "rank": 0, "rccl_version": 22605}

    def comm_destroy(self):
        pass


def test_collective_evidence_never_sinks_the_measurement(monkeypatch):
    """The evidence leg of --gpus N (the library's own RCCL communicator, first spanned over several GPUs on the driver's node) runs on a helper thread with a time
    limit: success fills `comm_ranks`; an exception or a hang becomes a WARNING on the line -- not an error, not a crash, not a wait."""
    import time
    monkeypatch.setattr(bench, "WARNINGS", [])
    monkeypatch.setattr(bench, "HUNG_THREAD", [])
    ev = bench.collective_evidence(_FakeCtx("ok"), _FakeDist(), 0, 1)
    assert ev["comm_ranks"] == 1 and ev["rccl_version"] == 22605 and not bench.WARNINGS
    ev = bench.collective_evidence(_FakeCtx("raise"), _FakeDist(), 0, 1)
    assert "ncclCommInitRank failed" in ev["error"] and ev["comm_ranks"] is None and len(bench.WARNINGS) == 1 and not bench.ERRORS
    t0 = time.time()
    ev = bench.collective_evidence(_FakeCtx("hang"), _FakeDist(), 0, 1, timeout_s=0.5)
    assert time.time() - t0 < 5 and "did not come up" in ev["error"] and len(bench.HUNG_THREAD) == 1
