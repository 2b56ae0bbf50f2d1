set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate"
python bench.py --config 4 --steps 3 $Q > $O/bench_c4.json 2>/dev/null
python bench.py --config 4 --steps 3 --sub-images -1 $Q > $O/bench_c4_onepair.json 2>/dev/null
python bench.py --config 4 --steps 20 --images 8 $Q > $O/bench_c4_8images.json 2>/dev/null
python bench.py --config 5 --steps 5 --single-chain --no-gather $Q 2>/dev/null | tail -1 > $O/bench_c5_single_chain.json
python bench.py --config 5 --steps 5 $Q > $O/bench_c5.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_c*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("instruction_floor"), d["roofline"].get("pmc_key"), d["roofline"].get("traffic"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
