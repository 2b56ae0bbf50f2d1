# round 4: evidence of the final build after the last two kernel changes (records requested early, static first ticket): default line, config 4 lines, 4096^2, kernel trace + PMC of the default
set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate"
bash tools/prof.sh r04_final --steps 25 | tail -1
bash tools/prof.sh r04_final_c4 --config 4 --steps 2 --warmup 1 | tail -1
bash tools/prof.sh r04_final_rg4096 --size 4096 --workload random_gradient | tail -1
cp gpurun_out/profiles/pmc_by_workload.json profiles/pmc_by_workload.json
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config 4 --steps 3 $Q > $O/bench_c4.json 2>/dev/null
python bench.py --config 4 --steps 3 --sub-images -1 $Q > $O/bench_c4_onepair.json 2>/dev/null
python bench.py --config 4 --steps 20 --images 8 $Q > $O/bench_c4_8images.json 2>/dev/null
python bench.py --size 4096 --workload random_gradient $Q > $O/bench_rg4096.json 2>/dev/null
python bench.py --config 5 --steps 5 $Q > $O/bench_c5.json 2>/dev/null
python bench.py --rgb $Q > $O/bench_rgb.json 2>/dev/null
python bench.py --accurate --steps 10 $Q > $O/bench_accurate.json 2>/dev/null
python bench.py --error-factor 25 $Q > $O/bench_ef25.json 2>/dev/null
python bench.py --error-factor 400 $Q > $O/bench_ef400.json 2>/dev/null
python bench.py --float-mode fast $Q > $O/bench_fast.json 2>/dev/null
python bench.py --steps 20 --size 8192x8190 $Q > $O/bench_8192x8190.json 2>/dev/null
python bench.py --stream $Q > $O/bench_stream.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list((d["roofline"].get("kernels_ms") or {}).values()), d["roofline"].get("frac"), d["roofline"].get("instruction_floor"))
    except Exception as e: pass
PY
