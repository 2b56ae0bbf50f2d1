set -o pipefail
O=gpurun_out/r04_7; mkdir -p $O
for cfg in "1000000000 1 0" "1000000000 1 1" "8192 4 1" "16384 2 1" "32768 2 1" "32768 1 1" "65536 1 0"; do
  set -- $cfg
  tag=b$1_s$2_p$3
  LIMG_BLK_BATCH=$1 LIMG_BLK_STREAMS=$2 LIMG_BLK_PRIO=$3 LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 5 --no-cpu-baseline > $O/pn_$tag.json 2>$O/pn_$tag.err
  LIMG_BLK_BATCH=$1 LIMG_BLK_STREAMS=$2 LIMG_BLK_PRIO=$3 LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 5 --no-cpu-baseline --workload random_gradient > $O/rg_$tag.json 2>$O/rg_$tag.err
  echo $tag; tail -1 $O/pn_$tag.err; tail -1 $O/rg_$tag.err
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("stage_ms"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
