# round 4 final: the bench line of every mode, against this round's counters (profiles/pmc_by_workload.json must be the round's).  usage: bash tools/r04/suite_bench.sh <part>
set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate"
case "$1" in
 1) python bench.py > $O/bench.json 2> $O/bench.err
    python bench.py --float-mode fast $Q > $O/bench_fast.json 2>/dev/null
    python bench.py --split $Q > $O/bench_split.json 2>/dev/null
    python bench.py --legacy-float-stage $Q > $O/bench_legacy.json 2>/dev/null
    python bench.py --accurate --steps 10 $Q > $O/bench_accurate.json 2>/dev/null
    python bench.py --rgb $Q > $O/bench_rgb.json 2>/dev/null
    python bench.py --error-factor 25 $Q > $O/bench_ef25.json 2>/dev/null
    python bench.py --error-factor 400 $Q > $O/bench_ef400.json 2>/dev/null
    python bench.py --size 4096 --workload random_gradient $Q > $O/bench_rg4096.json 2>/dev/null ;;
 2) python bench.py --config 4 --steps 3 $Q > $O/bench_c4.json 2>/dev/null
    python bench.py --config 4 --steps 3 --sub-images -1 $Q > $O/bench_c4_onepair.json 2>/dev/null
    python bench.py --config 4 --steps 3 --no-batch $Q > $O/bench_c4_nobatch.json 2>/dev/null
    python bench.py --config 4 --steps 3 --no-batch --contexts 3 $Q > $O/bench_c4_contexts3.json 2>/dev/null
    python bench.py --config 4 --steps 20 --images 8 $Q > $O/bench_c4_8images.json 2>/dev/null
    python bench.py --config 5 --steps 5 $Q > $O/bench_c5.json 2>/dev/null
    python bench.py --config 5 --steps 5 --single-chain --no-gather $Q 2>/dev/null | tail -1 > $O/bench_c5_single_chain.json
    python bench.py --stream $Q > $O/bench_stream.json 2>/dev/null
    python bench.py --blocked --steps 5 --contexts 4 > $O/bench_blocked.json 2>/dev/null
    python bench.py --blocked --steps 5 --workload random_gradient --no-cpu-baseline > $O/bench_blocked_rg.json 2>/dev/null
    python bench.py --steps 20 --size 8192x8190 $Q > $O/bench_8192x8190.json 2>/dev/null
    python bench.py --steps 5 --size 8192x8190 --whole-image-ragged $Q > $O/bench_8192x8190_whole_image_path.json 2>/dev/null
    python bench.py --steps 5 --size 8190x8192 $Q > $O/bench_8190x8192.json 2>/dev/null
    python bench.py --steps 20 --size 1024x618 --rgb $Q > $O/bench_1024x618_rgb.json 2>/dev/null
    python bench.py --steps 20 --size 1024x618 --rgb --whole-image-ragged $Q > $O/bench_1024x618_rgb_whole_image_path.json 2>/dev/null
    python bench.py --steps 20 --size 1024x616 --rgb $Q > $O/bench_1024x616_rgb.json 2>/dev/null ;;
esac
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"), d["roofline"].get("instruction_floor"), d["roofline"].get("pmc_refused_stale_source"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
