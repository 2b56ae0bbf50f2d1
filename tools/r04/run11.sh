set -o pipefail
O=gpurun_out/r04_11; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
for w in 8 16 2; do mkdir -p /tmp/ab_w$w; python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_TPB_WG_WAVES=$w'], out_dir='/tmp/ab_w$w')" > $O/build_w$w.log 2>&1; done
for rep in 1 2; do for w in 4 8 16 2; do
  L=limg_amd/liblimg_hip.so; [ $w != 4 ] && L=/tmp/ab_w$w/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py $B --steps 30 --size 4096 --workload random_gradient > $O/rg4096_w${w}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 30 > $O/pn8192_w${w}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
