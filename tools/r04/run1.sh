# round 4, first GPU session: new tests, then A/B of the sub-batch pipeline (config 4) and the ragged paths.  usage: bash tools/r04/run1.sh
set -o pipefail
O=gpurun_out/r04_1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_collective.py tests/test_shim_threads.py tests/test_stats.py tests/test_gpu_stream.py -m gpu -q -x > $O/new_tests.log 2>&1; tail -3 $O/new_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ragged or hashes or golden" > $O/ragged_tests.log 2>&1; tail -3 $O/ragged_tests.log
B="--no-cpu-baseline --no-host-rate"
for sub in 0 2 4 8 16; do
  python bench.py --config 4 --steps 3 $B --sub-images $sub > $O/c4_sub$sub.json 2>$O/c4_sub$sub.err
done
python bench.py --config 4 --steps 3 $B --wg-per-cu 5 > $O/c4_wg5.json 2>/dev/null
python bench.py --config 4 --steps 3 $B --workload photo_noise --size 4096 > $O/c4_pn_sub0.json 2>/dev/null
python bench.py --config 4 --steps 3 $B --workload photo_noise --size 4096 --sub-images 8 > $O/c4_pn_sub8.json 2>/dev/null
python bench.py $B --steps 20 > $O/b8192.json 2>$O/b8192.err
python bench.py $B --steps 20 --size 8192x8190 > $O/b8192x8190.json 2>$O/b8192x8190.err
python bench.py $B --steps 5 --size 8192x8190 --whole-image-ragged > $O/b8192x8190_whole.json 2>$O/b8192x8190_whole.err
python bench.py $B --steps 5 --size 8190x8192 > $O/b8190x8192.json 2>$O/b8190x8192.err
python bench.py $B --steps 20 --size 1024x618 --rgb > $O/b1024x618.json 2>$O/b1024x618.err
python bench.py $B --steps 20 --size 1024x618 --rgb --whole-image-ragged > $O/b1024x618_whole.json 2>/dev/null
python bench.py $B --steps 20 --size 1024x616 --rgb > $O/b1024x616.json 2>/dev/null
python bench.py $B --steps 20 --size 4096 --workload random_gradient > $O/rg4096.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
