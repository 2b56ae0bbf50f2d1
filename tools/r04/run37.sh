#!/bin/bash
# the merge's look-ahead with the centre rows (now the default): merged-block tests, fuzz, bench lines (photo-noise / gradient, contexts 1 / 4 / 8)
set -o pipefail
O=gpurun_out/r37; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py --seconds 120 --seed 701 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python bench.py --blocked --steps 8 --warmup 2 --contexts 4 > $O/bench_blocked.json 2>/dev/null
python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --workload random_gradient --no-cpu-baseline > $O/bench_blocked_rg.json 2>/dev/null
python bench.py --blocked --steps 8 --warmup 2 --contexts 8 --no-cpu-baseline > $O/pn_c8.json 2>/dev/null
python bench.py --blocked --steps 8 --warmup 2 --contexts 8 --workload random_gradient --no-cpu-baseline > $O/rg_c8.json 2>/dev/null
LIMG_HIP_DEBUG_TIMELINE=1 python bench.py --blocked --steps 1 --warmup 2 --no-cpu-baseline > $O/tl.json 2> $O/tl.err; grep "^batch\|^merge" $O/tl.err | tail -30 > $O/timeline.txt; tail -3 $O/timeline.txt; LIMG_HIP_DEBUG_TIMELINE=1 python bench.py --blocked --steps 1 --warmup 2 --no-cpu-baseline --workload random_gradient > $O/tlrg.json 2> $O/tlrg.err; grep "^batch\|^merge" $O/tlrg.err | tail -40 > $O/timeline_rg.txt
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); st = d["config"]["stage_ms"]; p = d["config"].get("pipelined_stream") or {}
    print(os.path.basename(f), d["ms_per_step"], {k: st[k] for k in ("pass1_match_gpu", "merge_host", "chain_host", "total")}, "contexts", p.get("contexts"), p.get("Mpixels_per_s"))
PY
