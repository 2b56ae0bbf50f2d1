#!/usr/bin/env bash
# kernel trace of the stream path's kernels (tools/r06/time_pack.py) on the GPU box -> gpurun_out/$1/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/r06/time_pack.py > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1] + "/trace/trace_kernel_trace.csv"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the image size changes after the first 24 stream encodes: report per half
names = ("k_stream_scan_strips", "k_stream_pack_strips", "k_stream_count", "k_stream_scan(", "k_stream_pack(", "k_stream_decode")
for nm in names:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if nm in r["Kernel_Name"]]
    if d:
        h = len(d) // 2
        print("%-24s n=%d  first half avg %.1f us (min %.1f)   second half avg %.1f us (min %.1f)" % (nm, len(d), sum(d[:h]) / h / 1e3, min(d[:h]) / 1e3, sum(d[h:]) / (len(d) - h) / 1e3, min(d[h:]) / 1e3))
PY
