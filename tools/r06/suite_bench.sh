#!/usr/bin/env bash
# Round 6 evidence, part 2 (GPU box): the bench lines of every mode on the build in the tree -> gpurun_out/r06_final/bench_<mode>.json (bench.py stamps each with
# head / lib_sha / src_sha; a failed leg makes bench.py exit non-zero, which is recorded in rc.txt).  tools/r06/collect.sh copies them to profiles/r06_bench_final_<mode>.json.
O=gpurun_out/r06_final; mkdir -p $O; : > $O/rc.txt
Q="--no-cpu-baseline --no-host-rate"
run() { name=$1; shift; python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?" >> $O/rc.txt; }
run default
run accurate_rg4096 --accurate --size 4096 --workload random_gradient --steps 10 $Q
run host_pool2 --steps 5 --pool-threads 2 --no-cpu-baseline
run c4 --config 4 --steps 3 --verify-golden $Q
run c4_contexts3 --config 4 --steps 3 --contexts 3 $Q
run c4_8images --config 4 --steps 20 --images 8 $Q
run rg4096 --size 4096 --workload random_gradient $Q
run c5 --config 5 --steps 5 $Q
run c5_single_chain --config 5 --steps 5 --single-chain $Q
run rgb --rgb $Q
run accurate --accurate --steps 10 $Q
run ef25 --error-factor 25 $Q
run ef400 --error-factor 400 $Q
run fast --float-mode fast $Q
run legacy --legacy-float-stage --steps 20 $Q
run split --split --steps 20 $Q
run 8192x8190 --steps 20 --size 8192x8190 $Q
run 8190x8192 --steps 5 --warmup 1 --size 8190x8192 $Q
run 8190x8192_pool2 --steps 10 --warmup 1 --size 8190x8192 --pool-threads 2 $Q
run 8190x8192_ctx4 --steps 5 --warmup 1 --size 8190x8192 --contexts 4 $Q
run 1024x618 --steps 50 --size 1024x618 --rgb $Q
run gpus2_c5_single_chain_rehearsal --gpus 2 --config 5 --single-chain --share-gpus --verify-golden --no-gather --steps 2 --warmup 1 $Q
run gpus2_c4_rehearsal --gpus 2 --config 4 --images 16 --share-gpus --verify-golden --no-gather --steps 2 --warmup 1 $Q
run gpus2_default_rehearsal --gpus 2 --share-gpus --steps 5 --warmup 2 $Q
run gpus2_c5_rehearsal --gpus 2 --config 5 --share-gpus --verify-golden --no-gather --steps 2 --warmup 1 $Q
run stream --stream $Q
run blocked --blocked --steps 6 --contexts 4 $Q
run blocked_rg --blocked --steps 6 --contexts 4 --workload random_gradient $Q
cat $O/rc.txt
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list((d["roofline"].get("kernels_ms") or {}).values()), d["roofline"].get("frac"), "errors:", d.get("errors"))
    except Exception as e: print(os.path.basename(f), "UNREADABLE", e)
PY
