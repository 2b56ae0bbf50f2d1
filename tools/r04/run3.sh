# round 4, third GPU session: k_fit_tpb with prefetched rows against the round-3 form, same box.  usage: bash tools/r04/run3.sh
set -o pipefail
O=gpurun_out/r04_3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stagewise or hashes or golden or ragged" > $O/tests.log 2>&1; tail -2 $O/tests.log
B="--no-cpu-baseline --no-host-rate"
mkdir -p /tmp/ab_nopf /tmp/ab_nopf8 && python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_TPB_NO_PREFETCH=1'], out_dir='/tmp/ab_nopf'); build.build(force=True, extra_flags=['-DLIMG_TPB_NO_PREFETCH=1', '-DLIMG_TPB_WAVES_PER_SIMD=8'], out_dir='/tmp/ab_nopf8')" > $O/build.log 2>&1
for rep in 1 2; do
for v in new nopf nopf8; do
  L=limg_amd/liblimg_hip.so; [ $v = nopf ] && L=/tmp/ab_nopf/liblimg_hip.so; [ $v = nopf8 ] && L=/tmp/ab_nopf8/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py $B --steps 30 > $O/${v}_8192_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 30 --size 4096 --workload random_gradient > $O/${v}_rg4096_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 > $O/${v}_c4_sub0_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 --sub-images 8 > $O/${v}_c4_sub8_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 --sub-images 8 --pipeline-knobs 0x1 > $O/${v}_c4_sub8_p0_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
