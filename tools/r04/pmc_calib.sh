#!/usr/bin/env bash
# Dynamic instruction counts of k_encode_persistent on workloads whose search statistics differ (errorFactor sweep, search bypassed): the input of
# tools/isa_calibrate.py.  One rocprofv3 --pmc pass each (instruction counters only; never together with tracing).  usage: bash tools/r04/pmc_calib.sh
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_calib; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in "ef25 --error-factor 25" "ef50 --error-factor 50" "ef100 --error-factor 100" "ef200 --error-factor 200" "ef400 --error-factor 400" "shift0 --forced-shift 0" "shift4 --forced-shift 4"; do
  set -- $W
  tag=$1; shift
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/$tag" -o pmc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-host-rate "$@" > "$OUT/$tag.log" 2>&1 || { echo "$tag failed"; tail -3 "$OUT/$tag.log"; }
done
python3 "$R/tools/isa_calibrate.py" "$OUT" "$R/profiles/r04_search_stats.json" | tee "$OUT/calibration.md"
