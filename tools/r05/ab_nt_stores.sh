#!/usr/bin/env bash
# Same-box A/B: the output planes stored non-temporally (the build in the tree) against plain stores (ab/liblimg_hip_temporal.so = the same sources with
# -DLIMG_PLANE_STORES_TEMPORAL), on the headline image, config 2's image and config 4's list; then the counters of config 4 on the tree's build.
O=gpurun_out/r05_nt; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate"
for i in 1 2; do
  for v in nt temporal; do
    L=""; [ $v = temporal ] && L="$PWD/ab/liblimg_hip_temporal.so"
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
bash tools/prof.sh r05_nt_c4 --config 4 --steps 2 --warmup 1 | tail -1
bash tools/prof.sh r05_nt_default --steps 10 | tail -1
python - <<'PY'
import json
d = json.load(open("gpurun_out/profiles/pmc_by_workload.json"))
for key, px in (("config4_batched_n64_sub8", 8 * 4096 * 4096), ("8192x8192_photo_noise_ef100_fused", 8192 * 8192)):
    e = d[key]; print(key, e["source"], "valu_busy", e.get("valu_busy"))
    for k, v in e["per_kernel"].items():
        print("  %-22s fetch %.2f B/px  write %.2f B/px" % (k, 2 * v["fetch_kib"] * 1024 / px, v["write_kib"] * 1024 / px))
PY
