#!/bin/bash
# same-box A/B of the worker's batching policy (abtmp/<v>: first report of the merge / rectangles worth a launch with a batch in flight / from inside a walk / walk piece)
set -o pipefail
O=gpurun_out/r38; mkdir -p $O
for rep in 1 2 3; do for v in old b c d e; do
  L=abtmp/$v/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 8 --warmup 2 --no-cpu-baseline > $O/pn_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 8 --warmup 2 --no-cpu-baseline --workload random_gradient > $O/rg_${v}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = os.path.basename(f).rsplit("_", 1)[0]
    acc[k].append(d["ms_per_step"])
for k, v in sorted(acc.items()): print(k, v)
PY
