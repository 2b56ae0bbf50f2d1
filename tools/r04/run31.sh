#!/bin/bash
# merged-block encoder: images pipelined over 2 / 4 / 6 / 8 contexts on the last build
set -o pipefail
O=gpurun_out/r31; mkdir -p $O
for k in 2 4 6 8; do
  python bench.py --blocked --steps 8 --warmup 2 --contexts $k --no-cpu-baseline > $O/pn_c$k.json 2>/dev/null
  python bench.py --blocked --steps 8 --warmup 2 --contexts $k --workload random_gradient --no-cpu-baseline > $O/rg_c$k.json 2>/dev/null
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); p = d["config"].get("pipelined_stream") or {}
    print(os.path.basename(f), d["ms_per_step"], "contexts", p.get("contexts"), "Mpx/s", p.get("Mpixels_per_s"), "images/s", p.get("images_per_s"))
PY
