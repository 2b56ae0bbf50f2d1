# round 4: ragged paths with chain-state upload; merged-block encoder timeline.  usage: bash tools/r04/run5.sh
set -o pipefail
O=gpurun_out/r04_5; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_blocked.py tests/test_gpu_stream.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
B="--no-cpu-baseline --no-host-rate"
python bench.py $B --steps 5 --size 8190x8192 > $O/b8190x8192.json 2>$O/b8190x8192.err
python bench.py $B --steps 5 --size 8190x8190 > $O/b8190x8190.json 2>$O/b8190x8190.err
python bench.py $B --steps 20 --size 8192x8190 > $O/b8192x8190.json 2>$O/b8192x8190.err
python bench.py $B --steps 20 --size 1024x618 --rgb > $O/b1024x618.json 2>/dev/null
LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 4 --no-cpu-baseline > $O/blocked.json 2>$O/blocked.err
LIMG_HIP_DEBUG_TIMING=1 python bench.py --blocked --steps 4 --no-cpu-baseline --workload random_gradient > $O/blocked_rg.json 2>$O/blocked_rg.err
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"), d["config"].get("stage_ms"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
tail -4 $O/blocked.err; tail -4 $O/blocked_rg.err
