#!/usr/bin/env bash
# same-box A/B of the merged-block encoder with / without the device-side "large rectangles first" order (test build: the switch is a test hook)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06h}; mkdir -p $OUT
export LIMG_HIP_LIB=test
for rep in 1 2; do
  for o in "" "${AB_FLAG:---no-order}"; do
    for w in photo_noise random_gradient; do
      python bench.py --blocked --workload $w --steps 6 --warmup 2 --contexts 4 --no-cpu-baseline $o > $OUT/blk_${w}_$rep$o.log 2>&1
      python - $OUT/blk_${w}_$rep$o.log "$w ${o:-ordered}" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        s = d["config"]["stage_ms"]
        print("blocked %-28s %.2f ms/image  fit_search %.2f  expand+store %.2f  chain %.2f  merge %.2f | 4 contexts %.0f Mpx/s  errors %s"
              % (sys.argv[2], d["ms_per_step"], s["fit_search_kernel"], s["expand_store_kernels"], s["chain_host"], s["merge_host"], d["config"]["pipelined_stream"]["Mpixels_per_s"], d["errors"]))
PY
    done
  done
done
