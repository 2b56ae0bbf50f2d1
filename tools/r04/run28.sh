#!/bin/bash
# k_blocked_match at three workgroups per CU (now the default): merged-block tests, fuzz (blocked leg included), the bench lines
set -o pipefail
O=gpurun_out/r28; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py --seconds 120 --seed 301 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
for rep in 1 2 3; do
  python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --no-cpu-baseline > $O/pn_$rep.json 2>/dev/null
done
python bench.py --blocked --steps 8 --warmup 2 --contexts 4 > $O/bench_blocked.json 2>/dev/null
python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --workload random_gradient --no-cpu-baseline > $O/bench_blocked_rg.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); st = d["config"]["stage_ms"]; p = d["config"].get("pipelined_stream") or {}
    print(os.path.basename(f), d["ms_per_step"], "match", st["match_kernels"], "fit_search", st["fit_search_kernel"], "store", st["expand_store_kernels"], "c4", p.get("Mpixels_per_s"), d["roofline"]["frac"])
PY
