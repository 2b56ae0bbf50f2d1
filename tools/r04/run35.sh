#!/bin/bash
# merged-block encoder: what the merge publishes is enqueued between the pieces of a batch's chain walk (not when the walk is over)
set -o pipefail
O=gpurun_out/r35; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
LIMG_HIP_DEBUG_TIMELINE=1 python bench.py --blocked --steps 1 --warmup 2 --no-cpu-baseline > $O/tl.json 2> $O/tl.err; tail -16 $O/tl.err
timeout -k 10 200 python tools/fuzz_gpu.py --seconds 100 --seed 601 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
for rep in 1 2 3; do
  python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --no-cpu-baseline > $O/pn_$rep.json 2>/dev/null
  python bench.py --blocked --steps 8 --warmup 2 --contexts 4 --workload random_gradient --no-cpu-baseline > $O/rg_$rep.json 2>/dev/null
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); st = d["config"]["stage_ms"]; p = d["config"].get("pipelined_stream") or {}
    print(os.path.basename(f), d["ms_per_step"], {k: st[k] for k in ("pass1_match_gpu", "merge_host", "fit_search_gpu", "chain_host", "total")}, "c4", p.get("Mpixels_per_s"))
PY
