# round 4: merged-block encoder after the host-side changes (row-mask expansion, look-ahead prefetch, chain values instead of noise bytes, second stream)
set -o pipefail
O=gpurun_out/r04_6; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_blocked.py tests/test_cli.py tests/test_stats.py tests/test_shim_threads.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1; do
LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 6 --no-cpu-baseline > $O/blocked_$rep.json 2>$O/blocked_$rep.err
LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 6 --no-cpu-baseline --workload random_gradient > $O/blocked_rg_$rep.json 2>$O/blocked_rg_$rep.err
done
python bench.py --blocked --steps 6 --contexts 4 --no-cpu-baseline > $O/blocked_ctx4.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("stage_ms"), d["config"].get("pipelined_stream"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
tail -3 $O/blocked_1.err; tail -3 $O/blocked_rg_1.err
timeout -k 10 500 python tools/fuzz_gpu.py --seconds 120 --seed 41 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
