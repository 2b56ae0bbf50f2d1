#!/bin/bash
# merged-block encoder: the worker's timeline of one image (LIMG_HIP_DEBUG_TIMELINE)
set -o pipefail
O=gpurun_out/r34; mkdir -p $O
LIMG_HIP_DEBUG_TIMELINE=1 python bench.py --blocked --steps 2 --warmup 2 --no-cpu-baseline > $O/pn.json 2> $O/pn.err; tail -14 $O/pn.err
LIMG_HIP_DEBUG_TIMELINE=1 python bench.py --blocked --steps 2 --warmup 2 --no-cpu-baseline --workload random_gradient > $O/rg.json 2> $O/rg.err; tail -14 $O/rg.err
