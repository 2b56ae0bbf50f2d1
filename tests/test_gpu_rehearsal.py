"""Multi-GPU readiness on ONE card (no multi-GPU node is available to the builder): `bench.py --gpus 2 --config 5 --share-gpus` -- two rank PROCESSES, each with its
own context, sharing the GPU, torch.distributed over gloo -- at BASELINE config 5's real size (16384^2 photo-noise), every rank checking the strips it produced
against the REAL reference's per-strip checksums (tests/golden/fullsize.json, tools/make_golden_fullsize.py; reference: the strip scheduler of src/limg.cpp:2105-2138).
  * --single-chain: ONE dither chain through both ranks' halves (== pThreadPool == nullptr): rank 1's planes are only right if its chain base -- the count rank 0
    all-gathered -- is; on a multi-GPU node the same bench line runs through RCCL (limg_hip_encode3d_single_chain_device), here the 8 bytes travel over gloo and the
    entry's two exchange-free halves do the rest;
  * strip restart (the reference with a pool of 2 threads): no exchange at all.
The two ranks' kernels run on the GPU at the same time, which is safe since every strip id of the persistent kernel comes from the atomic ticket
(tests/test_gpu_concurrency.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--share-gpus", "--verify-golden", "--no-gather", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-host-rate"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[0])


def test_config5_single_chain_two_ranks_on_one_gpu():
    line = _run(["--single-chain"])
    assert line["errors"] == [] and line["n_gpus"] == 2
    c = line["config"]
    assert c["collective"]["comm_ranks"] == 2 and c["collective"]["world_size"] == 2
    assert c["golden"]["ok"] is True and c["golden"]["entry"] == "pn16384_strips"
    assert c["golden"]["strips_checked_by_rank"] == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_config5_strip_restart_two_ranks_on_one_gpu():
    line = _run([])
    assert line["errors"] == [] and line["n_gpus"] == 2
    c = line["config"]
    assert c["golden"]["ok"] is True and c["golden"]["entry"] == "pn16384_pool2_strips"
    assert c["golden"]["strips_checked_by_rank"] == [[0, 1, 2, 3], [4, 5, 6, 7]]
