# round 4, last GPU session of the final build: smoke, the whole GPU suite once more, the merged-block lines, a two-rank gloo rehearsal of bench.py --gpus 2 on one card
set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python -m pytest tests -m gpu -q -x > $O/full_gpu_last.log 2>&1; tail -2 $O/full_gpu_last.log
python bench.py --blocked --steps 5 --contexts 4 > $O/bench_blocked.json 2>/dev/null
python bench.py --blocked --steps 5 --workload random_gradient --no-cpu-baseline > $O/bench_blocked_rg.json 2>/dev/null
python bench.py --gpus 2 --share-gpus --steps 5 --warmup 2 --no-cpu-baseline --no-host-rate > $O/bench_gpus2_gloo_rehearsal.json 2>$O/bench_gpus2.err; tail -c 600 $O/bench_gpus2_gloo_rehearsal.json; echo
python bench.py --gpus 2 --share-gpus --config 4 --images 8 --steps 3 --no-cpu-baseline --no-host-rate > $O/bench_gpus2_c4_gloo_rehearsal.json 2>>$O/bench_gpus2.err; tail -c 300 $O/bench_gpus2_c4_gloo_rehearsal.json; echo
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_blocked*.json")) + sorted(glob.glob(sys.argv[1] + "/bench_gpus2*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["n_gpus"], d["config"].get("stage_ms"), d["config"].get("collective"), (d["config"].get("pipelined_stream") or {}).get("Mpixels_per_s"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
