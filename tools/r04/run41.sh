#!/bin/bash
# same-box A/B: the merge's scan for the next unused + flagged seed sixteen at a time (tree) against one branch per seed (abtmp/noscan, -DLIMG_MERGE_NO_SIMD_SCAN)
set -o pipefail
O=gpurun_out/r41; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2 3; do for v in simd noscan; do
  L=limg_amd/liblimg_hip.so; [ $v = simd ] || L=abtmp/$v/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 8 --warmup 2 --no-cpu-baseline > $O/pn_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 8 --warmup 2 --no-cpu-baseline --workload random_gradient > $O/rg_${v}_$rep.json 2>/dev/null
done; done
timeout -k 10 200 python tools/fuzz_gpu.py --seconds 100 --seed 1001 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python - "$O" <<'PY'
import json, sys, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = os.path.basename(f).rsplit("_", 1)[0]
    acc[k].append((d["ms_per_step"], d["config"]["stage_ms"]["merge_host"]))
for k, v in sorted(acc.items()): print(k, v)
PY
