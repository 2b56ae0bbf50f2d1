set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "accurate or golden or hashes or generic" > $O/acc_tests4.log 2>&1; tail -3 $O/acc_tests4.log
python bench.py --steps 30 --no-cpu-baseline --no-host-rate > $O/bench_default4.json 2> $O/bench_default4.err
python bench.py --accurate --steps 10 --no-cpu-baseline --no-host-rate > $O/bench_acc4.json 2> $O/bench_acc4.err
python - <<'PY'
import json
for n in ("default4","acc4"):
    try:
        d=json.load(open("gpurun_out/r03/bench_%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(n, "failed", e)
PY
