O=gpurun_out/r05_nt3; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate"
for i in 1 2; do
  for v in nt dectemp factemp bothtemp; do
    L=""; [ $v != nt ] && L="$PWD/ab/liblimg_hip_$v.so"
    LIMG_HIP_LIB=$L python bench.py --steps 30 $Q > $O/default_${v}_$i.json 2>/dev/null
    LIMG_HIP_LIB=$L python bench.py --config 4 --steps 3 $Q > $O/c4_${v}_$i.json 2>/dev/null
    LIMG_HIP_LIB=$L python bench.py --size 4096 --workload random_gradient $Q > $O/rg4096_${v}_$i.json 2>/dev/null
  done
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list((d["roofline"].get("kernels_ms") or {}).values()), d["roofline"].get("frac"), d.get("errors"))
    except Exception as e: print(os.path.basename(f), "UNREADABLE", e)
PY
