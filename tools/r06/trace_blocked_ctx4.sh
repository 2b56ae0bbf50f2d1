#!/usr/bin/env bash
# is the merged-block encoder at four contexts bound by the GPU or by the host?  kernel trace of bench.py --blocked --contexts 4: over the pipelined phase, the fraction of
# wall time with at least one kernel running, and the sum of kernel durations per wall second
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06k}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 bench.py --blocked --steps 4 --warmup 1 --contexts 4 --no-cpu-baseline > $OUT/trace.log 2>&1
python3 - $OUT/trace/trace_kernel_trace.csv <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t_end = rows[-1][1]
# the pipelined phase is the last stretch: take the last 60 % of the trace's time span
t0 = rows[0][0] + int(0.55 * (t_end - rows[0][0]))
sel = [(max(a, t0), b, n) for a, b, n in rows if b > t0]
wall = t_end - t0
busy, cur_a, cur_b = 0, None, None
for a, b, _ in sorted(sel):
    if cur_b is None or a > cur_b:
        if cur_b is not None: busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
tot = collections.Counter()
for a, b, n in sel:
    key = next((k for k in ("k_blocked_match", "k_blocked_fit_search", "k_blocked_store", "k_noise_expand_calls", "k_blocked_order", "k_fit_tpb", "k_blocked_bounds") if k in n), "other")
    tot[key] += b - a
print("window %.1f ms: some kernel running %.1f %% of it; sum of kernel durations = %.2f x wall" % (wall / 1e6, 100.0 * busy / wall, sum(tot.values()) / wall))
for k, v in tot.most_common():
    print("  %-22s %.2f x wall" % (k, v / wall))
PY
grep -o '"pipelined_stream": {[^}]*}' $OUT/trace.log
