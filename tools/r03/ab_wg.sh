O=gpurun_out/r03; mkdir -p $O
LIMG_HIP_LIB=ab/dp6/liblimg_hip.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x 2>&1 | tail -1
LIMG_HIP_LIB=ab/dp7/liblimg_hip.so LIMG_HIP_WG_PER_CU=7 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x 2>&1 | tail -1
for rep in 1 2; do
for V in "dp6 6" "dp7 7" "nodp6 6"; do
  set -- $V
  LIMG_HIP_LIB=ab/$1/liblimg_hip.so LIMG_HIP_WG_PER_CU=$2 python bench.py --steps 40 --no-cpu-baseline --no-host-rate > /tmp/ab_a.json 2>/dev/null
  LIMG_HIP_LIB=ab/$1/liblimg_hip.so LIMG_HIP_WG_PER_CU=$2 python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > /tmp/ab_c.json 2>/dev/null
  LIMG_HIP_LIB=ab/$1/liblimg_hip.so LIMG_HIP_WG_PER_CU=$2 python bench.py --accurate --steps 8 --no-cpu-baseline --no-host-rate > /tmp/ab_d.json 2>/dev/null
  python - "$1" <<'PY'
import json, sys
a = json.load(open('/tmp/ab_a.json')); c = json.load(open('/tmp/ab_c.json')); d = json.load(open('/tmp/ab_d.json'))
print(sys.argv[1], a["ms_per_step"], a["roofline"]["kernels_ms"], "c4", c["ms_per_step"], "acc", d["roofline"]["kernels_ms"]["k_encode_persistent"])
PY
done; done 2>&1 | tee $O/ab_wg.log
