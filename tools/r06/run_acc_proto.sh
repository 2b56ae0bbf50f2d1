#!/usr/bin/env bash
# the accurate-search prototype on the GPU box: timing + correctness, then counters (own pmc passes, never with tracing) -> gpurun_out/$1/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06_acc_proto}; mkdir -p $OUT
cd $R
python3 tools/r06/acc_proto.py --size 4096x1024 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_4096x1024.txt
python3 tools/r06/acc_proto.py --size 4096x4096 --reps 10 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_4096x4096.txt
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc$i -o pmc -- python3 tools/r06/acc_proto.py --size 4096x1024 --reps 4 > $OUT/pmc$i.log 2>&1)
done
cd $R && python3 tools/prof_summary.py $OUT | grep -A14 "k_acc" | tee $OUT/counters.txt
