set -o pipefail
O=gpurun_out/r04_15; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
for v in 0 2; do mkdir -p /tmp/ab_st$v; python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_STAGGER=$v'], out_dir='/tmp/ab_st$v')" > $O/build_$v.log 2>&1; done
for rep in 1 2 3; do for v in 1 0 2; do
  L=limg_amd/liblimg_hip.so; [ $v != 1 ] && L=/tmp/ab_st$v/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 --size 4096 --workload random_gradient > $O/rg4096_st${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 30 > $O/pn8192_st${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 --size 2048 > $O/pn2048_st${v}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
