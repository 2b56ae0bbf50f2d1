#!/usr/bin/env bash
# kernel trace of the merged-block encoder bench: tools/prof_blocked.sh <tag> [bench args]
set -uo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" --blocked --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
tail -n 1 "$OUT/trace.log"
python3 - "$OUT/trace/trace_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls=%-5s avg_us=%10.1f pct=%s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
