# NOTE: the LIMG_BLK_EXP hooks this script builds with were removed from limg_hip_blocked.hip together with the experiment (profiles/r04_blocked_pipeline.md); kept as the record of how the split was measured.
set -o pipefail
O=$PWD/gpurun_out/r04_13; mkdir -p $O
R=$PWD
for e in 1 2 3; do mkdir -p /tmp/ab_e$e; python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_BLK_EXP=$e'], out_dir='/tmp/ab_e$e')" > $O/build_e$e.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
for e in 0 1 2 3; do
  L=$R/limg_amd/liblimg_hip.so; [ $e != 0 ] && L=/tmp/ab_e$e/liblimg_hip.so
  LIMG_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/e$e -o t -- python3 $R/bench.py --blocked --steps 3 --warmup 1 --no-cpu-baseline > $O/e$e.log 2>&1
  python3 - $O/e$e <<'PY'
import csv, sys, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_search" in r["Name"]: print(sys.argv[1].split("/")[-1], r["Name"][:90], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "total_ms %.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
done
