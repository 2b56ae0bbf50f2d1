#!/bin/bash
# same-box A/B: the merge's small-rectangle pass on a second host thread beside the large one (--merge-threads 0 = automatic) against one after the other (1)
set -o pipefail
O=gpurun_out/r42; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2 3; do for mt in 1 0; do
  python bench.py --blocked --steps 8 --warmup 2 --merge-threads $mt --no-cpu-baseline > $O/pn_mt${mt}_$rep.json 2>/dev/null
  python bench.py --blocked --steps 8 --warmup 2 --merge-threads $mt --no-cpu-baseline --workload random_gradient > $O/rg_mt${mt}_$rep.json 2>/dev/null
done; done
for mt in 1 0; do
  python bench.py --blocked --steps 8 --warmup 2 --merge-threads $mt --contexts 4 --no-cpu-baseline > $O/pn_c4_mt${mt}_1.json 2>/dev/null
  python bench.py --blocked --steps 8 --warmup 2 --merge-threads $mt --contexts 4 --no-cpu-baseline --workload random_gradient > $O/rg_c4_mt${mt}_1.json 2>/dev/null
done
timeout -k 10 200 python tools/fuzz_gpu.py --seconds 100 --seed 1101 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
LIMG_HIP_DEBUG_TIMELINE=1 python bench.py --blocked --steps 1 --warmup 2 --no-cpu-baseline > $O/tl.json 2> $O/tl.err; grep "^batch\|^merge" $O/tl.err | tail -26 > $O/timeline.txt; tail -4 $O/timeline.txt
python - "$O" <<'PY'
import json, sys, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = os.path.basename(f).rsplit("_", 1)[0]; p = d["config"].get("pipelined_stream") or {}
    acc[k].append((d["ms_per_step"], d["config"]["stage_ms"]["merge_host"], p.get("Mpixels_per_s")))
for k, v in sorted(acc.items()): print(k, v)
PY
