set -o pipefail
O=gpurun_out/r04_8; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
for w in 0 5 4; do for rep in 1 2; do
python bench.py $B --steps 30 --size 4096 --workload random_gradient --wg-per-cu $w > $O/rg4096_wg${w}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
bash tools/r04/pmc_calib.sh
