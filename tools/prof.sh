#!/usr/bin/env bash
# Profile the bench on the GPU box: kernel trace + stats, then PMC passes (each in its own run, never with tracing).
# usage: tools/prof.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -uo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-host-rate $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || { echo "pmc$i failed ($SET)"; tail -3 "$OUT/pmc$i.log"; }
done
mkdir -p "$R/gpurun_out/profiles"
cp -n "$R/profiles/pmc_by_workload.json" "$R/gpurun_out/profiles/pmc_by_workload.json" 2>/dev/null || true
python3 "$R/tools/prof_summary.py" "$OUT" --update-json "$R/gpurun_out/profiles/pmc_by_workload.json" --kernel "${PROF_KERNEL:-k_encode_persistent,k_fit_tpb}" --source "prof_$TAG" | tee "$OUT/summary.txt"
