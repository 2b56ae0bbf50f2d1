#!/bin/bash
# fuzz on the round's last build (the batch leg now also varies the sub-batch pipeline, the fixed leg the whole-image ragged switch) + the merged-block tests
set -o pipefail
O=gpurun_out/r26; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -2 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 420 python tools/fuzz_gpu.py --seconds 360 --seed 204 > $O/fuzz_gpu_final_360s.log 2>&1 && tail -1 $O/fuzz_gpu_final_360s.log &&
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 330 --seed 205 --max-blocks 160 > $O/fuzz_gpu_final_large_330s.log 2>&1 && tail -1 $O/fuzz_gpu_final_large_330s.log
