set -o pipefail
O=$PWD/gpurun_out/r04_14; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/pmc$i -o pmc -- python3 $R/bench.py --blocked --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc$i.log 2>&1 || tail -3 $O/pmc$i.log
done
python3 - $O <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "blocked" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][-30:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s mean %.4g  sum %.4g (n=%d)" % (c, sum(v) / len(v), sum(v), len(v)))
PY
