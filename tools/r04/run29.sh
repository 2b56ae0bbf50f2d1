#!/bin/bash
# k_blocked_match with the certain-match bound in front of its 27-colour loop: bits == host evaluation (test), merged-block tests, fuzz, A/B against the kernel without it
set -o pipefail
O=gpurun_out/r29; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/fuzz_gpu.py --seconds 150 --seed 401 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
for rep in 1 2; do for v in bound nobound; do
  F=""; [ $v = nobound ] && F="--no-match-bound"
  python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --no-cpu-baseline $F > $O/pn_${v}_$rep.json 2>/dev/null
  python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --workload random_gradient --no-cpu-baseline $F > $O/rg_${v}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); st = d["config"]["stage_ms"]; p = d["config"].get("pipelined_stream") or {}
    print(os.path.basename(f), d["ms_per_step"], "match", st["match_kernels"], "fit_search", st["fit_search_kernel"], "store", st["expand_store_kernels"], "c4", p.get("Mpixels_per_s"), d["roofline"]["frac"])
PY
