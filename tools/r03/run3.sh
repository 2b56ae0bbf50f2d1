set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/full_gpu3.log 2>&1; tail -15 $O/full_gpu3.log
python bench.py --steps 30 > $O/bench_default3.json 2> $O/bench_default3.err
python bench.py --stream --steps 20 --no-cpu-baseline > $O/bench_stream3.json 2> $O/bench_stream3.err
python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > $O/bench_c4_batch3.json 2> $O/bench_c4_batch3.err
python bench.py --config 5 --steps 3 --single-chain --no-gather --no-cpu-baseline --no-host-rate > $O/bench_c5_chain3.json 2> $O/bench_c5_chain3.err
python - <<'PY'
import json
for n in ("default3","stream3","c4_batch3","c5_chain3"):
    try:
        d=json.load(open("gpurun_out/r03/bench_%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
        if n=="default3": print(d["config"].get("cold"), d["config"].get("first_encode_ms"), json.dumps(d.get("cpu_baseline"))[:900])
    except Exception as e: print(n, "failed", e)
PY
timeout -k 10 500 python tools/fuzz_gpu.py --seconds 430 --seed 31 > $O/fuzz3.log 2>&1; tail -2 $O/fuzz3.log
