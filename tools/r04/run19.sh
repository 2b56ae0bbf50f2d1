# NOTE: A/B of the early record loads against the round-3 order; the -D hook was removed once the result was in; kept as the record of the measurement.
set -o pipefail
O=gpurun_out/r04_19; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
mkdir -p /tmp/ab_old; python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_EARLY_RECORDS=0'], out_dir='/tmp/ab_old')" > $O/build.log 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2 3; do for v in new old; do
  L=limg_amd/liblimg_hip.so; [ $v = old ] && L=/tmp/ab_old/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 > $O/pn8192_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 --size 4096 --workload random_gradient > $O/rg4096_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 > $O/c4_${v}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list(d["roofline"].get("kernels_ms").values()), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
