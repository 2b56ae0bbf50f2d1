O=gpurun_out/r05_bfs_split; mkdir -p $O
for v in base nosearch nowalk nofit nofit_nosearch; do
  L=""; [ $v != base ] && L="$PWD/ab/liblimg_hip_$v.so"
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 4 --no-cpu-baseline > $O/pn_$v.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py --blocked --steps 4 --no-cpu-baseline --workload random_gradient > $O/rg_$v.json 2>/dev/null
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], list(d["roofline"]["kernels_ms"].values()))
    except Exception as e: print(os.path.basename(f), "UNREADABLE", e)
PY
