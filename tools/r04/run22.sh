# same-box A/B: the tree's library against abtmp/liblimg_hip_old.so (the previous commit's build, brought along as a file)
set -o pipefail
O=gpurun_out/r04_22; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_stream.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2 3; do for v in new old; do
  L=limg_amd/liblimg_hip.so; [ $v = old ] && L=abtmp/liblimg_hip_old.so
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 > $O/pn8192_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 --size 4096 --workload random_gradient > $O/rg4096_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 > $O/c4_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 20 --split > $O/split_${v}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list(d["roofline"].get("kernels_ms").values()), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
