#!/usr/bin/env bash
# steady-state four-context throughput of the merged-block encoder (8+ images per context), product settings against the round's two switches off (test build)
export LIMG_HIP_LIB=test
for rep in 1 2; do
  for o in "" "--no-order" "--no-vec-store" "--no-order --no-vec-store"; do
    python bench.py --blocked --steps 12 --warmup 2 --contexts 4 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); ps=d['config']['pipelined_stream']; print('%-28s one image %.2f ms | 4 contexts, %d images: %.0f Mpx/s' % ('$o' or 'product', d['ms_per_step'], ps['images'], ps['Mpixels_per_s']))"
  done
done
