#!/bin/bash
# same-box A/B of k_blocked_match variants (abtmp/<v>/liblimg_hip.so built with -DLIMG_MATCH_WGS=n [-DLIMG_MATCH_ROLL_X]): HIP-event kernel times of bench.py --blocked
set -o pipefail
O=gpurun_out/r27; mkdir -p $O
for rep in 1 2; do for v in base w2 w3 w3r w4r; do
  L=limg_amd/liblimg_hip.so; [ $v = base ] || L=abtmp/$v/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 6 --warmup 2 --no-cpu-baseline > $O/${v}_$rep.json 2>$O/${v}_$rep.err || { tail -3 $O/${v}_$rep.err; exit 1; }
done; done
LIMG_HIP_LIB=abtmp/w3/liblimg_hip.so python -m pytest tests/test_gpu_blocked.py -x -q -m gpu 2>&1 | tail -1
for v in base w3 w3r; do
  L=limg_amd/liblimg_hip.so; [ $v = base ] || L=abtmp/$v/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 6 --warmup 2 --contexts 4 --no-cpu-baseline > $O/${v}_c4.json 2>/dev/null
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); st = d["config"]["stage_ms"]; p = d["config"].get("pipelined_stream") or {}
    print(os.path.basename(f), d["ms_per_step"], "match", st["match_kernels"], "fit_search", st["fit_search_kernel"], "store", st["expand_store_kernels"], "c4", p.get("Mpixels_per_s"))
PY
