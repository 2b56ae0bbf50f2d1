#!/bin/bash
# long fuzz on the round's last library build (all four legs; small and large shapes)
set -o pipefail
O=gpurun_out/r39; mkdir -p $O
timeout -k 10 520 python tools/fuzz_gpu.py --seconds 460 --seed 901 > $O/fuzz_gpu_last_460s.log 2>&1 && tail -1 $O/fuzz_gpu_last_460s.log &&
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 340 --seed 902 --max-blocks 160 > $O/fuzz_gpu_last_large_340s.log 2>&1 && tail -1 $O/fuzz_gpu_last_large_340s.log
