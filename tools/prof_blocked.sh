#!/usr/bin/env bash
# The merged-block encoder under the profiler: kernel trace, then PMC passes (each in its own run, never with tracing), condensed PER IMAGE -- its kernels run in
# bands / batches, so per-dispatch means say little -- by tools/prof_blocked_summary.py into gpurun_out/profiles/pmc_by_workload.json[blocked_<W>x<H>_<workload>].
# usage: tools/prof_blocked.sh <tag> [bench args]      outputs under gpurun_out/prof_<tag>/
set -uo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=3; WARM=1
ARGS="--blocked --steps $STEPS --warmup $WARM --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || { echo "pmc$i failed ($SET)"; tail -3 "$OUT/pmc$i.log"; }
done
mkdir -p "$R/gpurun_out/profiles"
cp -n "$R/profiles/pmc_by_workload.json" "$R/gpurun_out/profiles/pmc_by_workload.json" 2>/dev/null || true
python3 "$R/tools/prof_blocked_summary.py" "$OUT" $((STEPS + WARM)) --update-json "$R/gpurun_out/profiles/pmc_by_workload.json" --source "prof_$TAG" | tee "$OUT/summary.txt"
