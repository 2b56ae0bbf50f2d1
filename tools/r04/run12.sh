set -o pipefail
O=$PWD/gpurun_out/r04_12; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for fs in -1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/fs$fs -o t -- python3 $R/bench.py --blocked --steps 3 --warmup 1 --no-cpu-baseline --forced-shift $fs > $O/fs$fs.log 2>&1
  python3 - $O/fs$fs <<'PY'
import csv, sys, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blocked" in r["Name"] or "noise_expand" in r["Name"]: print(sys.argv[1].split("/")[-1], r["Name"][:90], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "total_ms %.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
done
