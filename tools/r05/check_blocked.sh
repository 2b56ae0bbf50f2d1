#!/usr/bin/env bash
# merged-block encoder after a kernel change: parity (tests, fuzz), then same-box A/B of the bench line against another build (LIMG_AB_LIB)
O=gpurun_out/r05_blocked; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py tests/test_gpu_fullsize.py -x -q -m gpu -k "blocked or stagewise or window or device_entry or match" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
python tools/fuzz_gpu.py --seconds 90 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
for i in 1 2; do
  for v in new old; do
    L=""; [ $v = old ] && L="${LIMG_AB_LIB:-}"
    [ $v = old ] && [ -z "$L" ] && continue
    LIMG_HIP_LIB=$L python bench.py --blocked --steps 6 --contexts 4 --no-cpu-baseline > $O/pn_${v}_$i.json 2>/dev/null
    LIMG_HIP_LIB=$L python bench.py --blocked --steps 6 --contexts 4 --no-cpu-baseline --workload random_gradient > $O/rg_${v}_$i.json 2>/dev/null
  done
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], list(d["roofline"]["kernels_ms"].values()), (d["config"].get("pipelined_stream") or {}).get("Mpixels_per_s"), d["config"]["stage_ms"]["fit_search_gpu"], d.get("errors"))
    except Exception as e: print(os.path.basename(f), "UNREADABLE", e)
PY
