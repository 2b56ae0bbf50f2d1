set -o pipefail
O=gpurun_out/r04_21; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2 3; do
  python bench.py $B --steps 40 > $O/pn8192_$rep.json 2>/dev/null
  python bench.py $B --steps 40 --size 4096 --workload random_gradient > $O/rg4096_$rep.json 2>/dev/null
  python bench.py $B --steps 40 --size 2048 > $O/pn2048_$rep.json 2>/dev/null
  python bench.py $B --config 4 --steps 3 > $O/c4_$rep.json 2>/dev/null
done
timeout -k 10 300 python tools/fuzz_gpu.py --seconds 150 --seed 99 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list(d["roofline"].get("kernels_ms").values()), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
