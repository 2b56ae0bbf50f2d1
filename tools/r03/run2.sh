set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/full_gpu2.log 2>&1; tail -15 $O/full_gpu2.log
python bench.py --steps 30 --no-cpu-baseline --no-host-rate > $O/bench_default2.json 2> $O/bench_default2.err
python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > $O/bench_c4_batch2.json 2> $O/bench_c4_batch2.err
python bench.py --size 4096 --workload random_gradient --no-cpu-baseline --no-host-rate > $O/bench_rg4096_2.json 2> $O/bench_rg4096_2.err
python bench.py --split --steps 20 --no-cpu-baseline --no-host-rate > $O/bench_split2.json 2> $O/bench_split2.err
python - <<'PY'
import json
for n in ("default2","c4_batch2","rg4096_2","split2"):
    try:
        d=json.load(open("gpurun_out/r03/bench_%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(n, "failed", e)
PY
