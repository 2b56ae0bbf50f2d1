set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/quick_gpu.log 2>&1; tail -2 $O/quick_gpu.log
for i in 1 2; do python bench.py --steps 40 --no-cpu-baseline --no-host-rate > $O/quick_bench$i.json 2>/dev/null; done
python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > $O/quick_c4.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench1","bench2","c4"):
    d=json.load(open("gpurun_out/r03/quick_%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
PY
