set -o pipefail
O=$PWD/gpurun_out/r04_9; mkdir -p $O
mkdir -p /tmp/ab_skip2 && python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_MATCH_SKIP2=1'], out_dir='/tmp/ab_skip2')" > $O/build.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in base skip2; do for w in photo_noise random_gradient; do
  L=$R/limg_amd/liblimg_hip.so; [ $v = skip2 ] && L=/tmp/ab_skip2/liblimg_hip.so
  LIMG_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_$w -o t -- python3 $R/bench.py --blocked --steps 3 --warmup 1 --no-cpu-baseline --workload $w > $O/${v}_$w.log 2>&1
  python3 - $O/${v}_$w <<'PY'
import csv, sys, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blocked" in r["Name"] or "noise_expand" in r["Name"]: print(sys.argv[1].split("/")[-1], r["Name"][:70], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "total_ms %.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
done; done
