set -o pipefail
O=gpurun_out/r04_17; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocked.py tests/test_gpu_stream.py tests/test_cli.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
B="--no-cpu-baseline --no-host-rate"
python bench.py $B --steps 5 --size 8190x8192 > $O/b8190x8192.json 2>/dev/null
python bench.py $B --steps 20 --size 8192x8190 > $O/b8192x8190.json 2>/dev/null
python bench.py $B --steps 20 --size 1024x618 --rgb > $O/b1024x618.json 2>/dev/null
LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 6 --contexts 4 --no-cpu-baseline > $O/blocked.json 2>$O/blocked.err
LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 6 --no-cpu-baseline --workload random_gradient > $O/blocked_rg.json 2>$O/blocked_rg.err
timeout -k 10 300 python tools/fuzz_gpu.py --seconds 150 --seed 77 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["config"].get("stage_ms"), (d["config"].get("pipelined_stream") or {}).get("Mpixels_per_s"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
tail -2 $O/blocked.err
