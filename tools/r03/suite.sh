# final measurement suite of a build: tests, fuzz, bench lines of every mode, rocprofv3 kernel traces + PMC passes.  usage: bash tools/r03/suite.sh <tag>
set -o pipefail
T=${1:-r03_final}
O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/full_gpu.log 2>&1; tail -2 $O/full_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --float-mode fast --no-cpu-baseline --no-host-rate > $O/bench_fast.json 2>/dev/null
python bench.py --split --no-cpu-baseline --no-host-rate > $O/bench_split.json 2>/dev/null
python bench.py --legacy-float-stage --no-cpu-baseline --no-host-rate > $O/bench_legacy.json 2>/dev/null
python bench.py --accurate --steps 10 --no-cpu-baseline --no-host-rate > $O/bench_accurate.json 2>/dev/null
python bench.py --rgb --no-cpu-baseline --no-host-rate > $O/bench_rgb.json 2>/dev/null
python bench.py --error-factor 25 --no-cpu-baseline --no-host-rate > $O/bench_ef25.json 2>/dev/null
python bench.py --error-factor 400 --no-cpu-baseline --no-host-rate > $O/bench_ef400.json 2>/dev/null
python bench.py --size 4096 --workload random_gradient --no-cpu-baseline --no-host-rate > $O/bench_rg4096.json 2>/dev/null
python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > $O/bench_c4.json 2>/dev/null
python bench.py --config 4 --steps 3 --no-batch --no-cpu-baseline --no-host-rate > $O/bench_c4_nobatch.json 2>/dev/null
python bench.py --config 4 --steps 3 --no-batch --contexts 3 --no-cpu-baseline --no-host-rate > $O/bench_c4_contexts3.json 2>/dev/null
python bench.py --config 5 --steps 5 --no-cpu-baseline --no-host-rate > $O/bench_c5.json 2>/dev/null
python bench.py --config 5 --steps 5 --single-chain --no-gather --no-cpu-baseline --no-host-rate 2>/dev/null | tail -1 > $O/bench_c5_single_chain.json
python bench.py --stream --no-cpu-baseline --no-host-rate > $O/bench_stream.json 2>/dev/null
python bench.py --blocked --steps 5 --contexts 4 > $O/bench_blocked.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
bash tools/prof.sh ${T} --steps 25 > $O/prof.log 2>&1; tail -1 $O/prof.log
PROF_KERNEL=k_fit_search,k_dither_store,k_fit_tpb bash tools/prof.sh ${T}_split --split > $O/prof_split.log 2>&1; tail -1 $O/prof_split.log
bash tools/prof.sh ${T}_accurate --accurate > $O/prof_acc.log 2>&1; tail -1 $O/prof_acc.log
bash tools/prof.sh ${T}_rg4096 --size 4096 --workload random_gradient > $O/prof_rg.log 2>&1; tail -1 $O/prof_rg.log
PROF_KERNEL=k_stream_decode bash tools/prof.sh ${T}_stream --stream > $O/prof_stream.log 2>&1; tail -1 $O/prof_stream.log
bash tools/prof.sh ${T}_c4 --config 4 --steps 2 --warmup 1 > $O/prof_c4.log 2>&1; tail -1 $O/prof_c4.log
timeout -k 10 460 python tools/fuzz_gpu.py --seconds 400 --seed 97 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
