set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests/test_gpu_batch.py -q -x > $O/batch_tests.log 2>&1; tail -3 $O/batch_tests.log
python -m pytest tests -m gpu -q -x > $O/full_gpu1.log 2>&1; tail -3 $O/full_gpu1.log
python bench.py --steps 30 --no-cpu-baseline --no-host-rate > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > $O/bench_c4_batch.json 2> $O/bench_c4_batch.err
python bench.py --config 4 --steps 3 --no-batch --no-cpu-baseline --no-host-rate > $O/bench_c4_nobatch.json 2> $O/bench_c4_nobatch.err
python bench.py --config 4 --steps 3 --no-batch --contexts 3 --no-cpu-baseline --no-host-rate > $O/bench_c4_ctx3.json 2> $O/bench_c4_ctx3.err
python bench.py --config 4 --steps 3 --images 8 --no-cpu-baseline --no-host-rate > $O/bench_c4_batch8.json 2> $O/bench_c4_batch8.err
python bench.py --size 4096 --workload random_gradient --no-cpu-baseline --no-host-rate > $O/bench_rg4096.json 2> $O/bench_rg4096.err
python - <<'PY'
import json
for n in ("default","c4_batch","c4_nobatch","c4_ctx3","c4_batch8","rg4096"):
    try:
        d=json.load(open("gpurun_out/r03/bench_%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(n, "failed", e)
PY
