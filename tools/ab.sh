#!/usr/bin/env bash
# A/B of library builds on ONE GPU box (box-to-box spread is +-2 %, so only same-box comparisons mean anything):
#   tools/ab.sh limg_amd/liblimg_hip.so build/other/liblimg_hip.so ...
# For every library, twice: the default bench (8192^2 photo-noise) and BASELINE config 4 (64 x 4096^2 gradient), through LIMG_HIP_LIB.
# Build a variant into its own directory with  python -c "from limg_amd import build; build.build(force=True, extra_flags=['-DX=1'], out_dir='/abs/path')".
for rep in 1 2; do
for L in "$@"; do
  LIMG_HIP_LIB=$L python bench.py --steps 50 --no-cpu-baseline --no-host-rate > /tmp/ab_a.json
  LIMG_HIP_LIB=$L python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > /tmp/ab_b.json
  python - "$L" <<'PY'
import json, sys
a = json.load(open('/tmp/ab_a.json')); b = json.load(open('/tmp/ab_b.json'))
print(sys.argv[1], a["ms_per_step"], a["roofline"]["kernels_ms"], "config4", b["ms_per_step"])
PY
done; done
