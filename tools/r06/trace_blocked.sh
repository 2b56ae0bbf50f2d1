#!/usr/bin/env bash
# rocprofv3 kernel trace of the merged-block encoder (test build) with the given extra bench flag: per-kernel total time per image -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R && LIMG_HIP_LIB=test rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --blocked --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
python3 - $OUT/trace/trace_kernel_stats.csv "$*" <<'PY'
import csv, sys
print("flags:", sys.argv[2] or "(none)")
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("k_blocked_store", "k_blocked_fit_search", "k_blocked_order", "k_noise_expand_calls", "k_blocked_match"):
        if k in n:
            print("  %-24s calls %-5s total %.3f ms per image (4 images)" % (k, r["Calls"], float(r["TotalDurationNs"]) / 4e6))
PY
