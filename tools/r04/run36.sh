#!/bin/bash
# same-box A/B: the merge's look-ahead also requests the similarity rows a centre expansion may start from (abtmp/pf2, pf4: offsets 1..2 / 1..4 in both directions)
set -o pipefail
O=gpurun_out/r36; mkdir -p $O
for rep in 1 2 3; do for v in base pf2 pf4; do
  L=limg_amd/liblimg_hip.so; [ $v = base ] || L=abtmp/$v/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 8 --warmup 2 --no-cpu-baseline > $O/${v}_$rep.json 2>/dev/null
done; done
LIMG_HIP_LIB=abtmp/pf4/liblimg_hip.so python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); st = d["config"]["stage_ms"]
    print(os.path.basename(f), d["ms_per_step"], {k: st[k] for k in ("merge_host", "chain_host", "total")})
PY
