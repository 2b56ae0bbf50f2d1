O=gpurun_out/r05_tpbocc; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate"
for v in base pad30 pad44 pad70; do
  L=""; [ $v != base ] && L="$PWD/ab/liblimg_hip_$v.so"
  LIMG_HIP_LIB=$L python bench.py --steps 30 $Q > $O/default_$v.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py --size 4096 --workload random_gradient $Q > $O/rg4096_$v.json 2>/dev/null
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list((d["roofline"].get("kernels_ms") or {}).values()), d["roofline"].get("frac"), d.get("errors"))
    except Exception as e: print(os.path.basename(f), "UNREADABLE", e)
PY
cd /tmp && export TMPDIR=/tmp
for v in base pad44 pad70; do
  L=""; [ $v != base ] && L="$GRAFT_REPO_ROOT/ab/liblimg_hip_$v.so"
  LIMG_HIP_LIB=$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$v -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $Q > /dev/null 2>&1
  python3 - $GRAFT_REPO_ROOT/$O/pmc_$v $v <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fit_tpb" in r["Kernel_Name"] or "k_encode_persistent" in r["Kernel_Name"]:
            k = "fit_tpb" if "k_fit_tpb" in r["Kernel_Name"] else "persistent"
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(2 * sum(v) / len(v) * 1024 / (8192 * 8192), 2) for k, v in acc.items()}, "B/px fetched")
PY
done
