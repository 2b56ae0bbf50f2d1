O=gpurun_out/r03; mkdir -p $O
for rep in 1 2; do
for V in "acc_nocache6 6" "acc_cache5 5" "acc_nocache5 5"; do
  set -- $V
  LIMG_HIP_LIB=ab/$1/liblimg_hip.so LIMG_HIP_WG_PER_CU=$2 python bench.py --accurate --steps 10 --no-cpu-baseline --no-host-rate > /tmp/ab.json 2>/dev/null
  python - "$1" <<'PY'
import json, sys
a = json.load(open('/tmp/ab.json')); print(sys.argv[1], a["ms_per_step"], a["roofline"]["kernels_ms"])
PY
done; done 2>&1 | tee $O/ab_acc.log
