# round 4: sub-batch size sweep for lists of 64 and of 8 images (config 4 on 1 and on 8 GPUs), fit priority 0.  usage: bash tools/r04/run4.sh
set -o pipefail
O=gpurun_out/r04_4; mkdir -p $O
B="--no-cpu-baseline --no-host-rate --config 4"
run() { python bench.py $B "${@:2}" > $O/$1.json 2>$O/$1.err; }
for rep in 1 2; do
run n64_sub0_$rep --steps 3
run n64_sub4_$rep --steps 3 --sub-images 4
run n64_sub8_$rep --steps 3 --sub-images 8
run n64_sub16_$rep --steps 3 --sub-images 16
run n64_sub8_f2_$rep --steps 3 --sub-images 8 --pipeline-knobs 0x201
run n64_sub8_f4_$rep --steps 3 --sub-images 8 --pipeline-knobs 0x401
run n64_sub8_w4_$rep --steps 3 --sub-images 8 --pipeline-knobs 0x41
run n8_sub0_$rep --steps 20 --images 8
run n8_sub1_$rep --steps 20 --images 8 --sub-images 1
run n8_sub2_$rep --steps 20 --images 8 --sub-images 2
run n8_sub4_$rep --steps 20 --images 8 --sub-images 4
run n8_sub2_f1_$rep --steps 20 --images 8 --sub-images 2 --pipeline-knobs 0x101
run n16_sub0_$rep --steps 10 --images 16
run n16_sub4_$rep --steps 10 --images 16 --sub-images 4
run n16_sub2_$rep --steps 10 --images 16 --sub-images 2
run n4_sub0_$rep --steps 30 --images 4
run n4_sub1_$rep --steps 30 --images 4 --sub-images 1
run n4_sub2_$rep --steps 30 --images 4 --sub-images 2
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
