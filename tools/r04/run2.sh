# round 4, second GPU session: knobs of the sub-batch pipeline (config 4), same box.  usage: bash tools/r04/run2.sh
set -o pipefail
O=gpurun_out/r04_2; mkdir -p $O
B="--no-cpu-baseline --no-host-rate --config 4 --steps 3"
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_shim_threads.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
run() { python bench.py $B "${@:2}" > $O/$1.json 2>$O/$1.err; }
run sub0
run sub8_p3_w5 --sub-images 8
run sub8_p0_w5 --sub-images 8 --pipeline-knobs 0x1
run sub8_p1_w5 --sub-images 8 --pipeline-knobs 0x2
run sub8_p2_w5 --sub-images 8 --pipeline-knobs 0x3
run sub8_p3_w4 --sub-images 8 --pipeline-knobs 0x44
run sub8_p0_w4 --sub-images 8 --pipeline-knobs 0x41
run sub8_p3_w5_f2 --sub-images 8 --pipeline-knobs 0x254
run sub8_p3_w4_f2 --sub-images 8 --pipeline-knobs 0x244
run sub4_p3_w5_f2 --sub-images 4 --pipeline-knobs 0x254
run sub16_p3_w5_f2 --sub-images 16 --pipeline-knobs 0x254
run sub16_p3_w4_f2 --sub-images 16 --pipeline-knobs 0x244
run sub0_again
# the persistent kernel without the E-step priority (short searches), built on the box
mkdir -p /tmp/ab_noprio && python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_PRIO_E=0'], out_dir='/tmp/ab_noprio')" > $O/build_noprio.log 2>&1
LIMG_HIP_LIB=/tmp/ab_noprio/liblimg_hip.so python bench.py $B > $O/noprio_sub0.json 2>/dev/null
LIMG_HIP_LIB=/tmp/ab_noprio/liblimg_hip.so python bench.py $B --sub-images 8 > $O/noprio_sub8_p3_w5.json 2>/dev/null
LIMG_HIP_LIB=/tmp/ab_noprio/liblimg_hip.so python bench.py $B --sub-images 8 --pipeline-knobs 0x254 > $O/noprio_sub8_p3_w5_f2.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
