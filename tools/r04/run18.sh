set -o pipefail
O=gpurun_out/r04_18; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
mkdir -p /tmp/ab_ta; python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DLIMG_TICKET_AHEAD=1'], out_dir='/tmp/ab_ta')" > $O/build.log 2>&1
LIMG_HIP_LIB=/tmp/ab_ta/liblimg_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stagewise or hashes or strip_restart or determinism" > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2; do for v in base ta; do
  L=limg_amd/liblimg_hip.so; [ $v = ta ] && L=/tmp/ab_ta/liblimg_hip.so
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 > $O/pn8192_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --steps 40 --size 4096 --workload random_gradient > $O/rg4096_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 > $O/c4_${v}_$rep.json 2>/dev/null
  LIMG_HIP_LIB=$L python bench.py $B --config 4 --steps 3 --sub-images -1 > $O/c4one_${v}_$rep.json 2>/dev/null
done; done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list(d["roofline"].get("kernels_ms").values()), d["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
