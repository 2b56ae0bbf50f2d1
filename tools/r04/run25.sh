#!/bin/bash
# the merged-block encoder's line with its kernel-only rate (HIP events of the worker's launches), the ragged / blocked tests on this build
set -o pipefail
O=gpurun_out/r25; mkdir -p $O
python -m pytest tests/test_gpu_blocked.py tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -2 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
python bench.py --blocked --steps 8 --warmup 2 --contexts 4 > $O/bench_blocked.json 2> $O/bench_blocked.err && cut -c1-1800 $O/bench_blocked.json &&
python bench.py --blocked --steps 8 --warmup 2 --workload random_gradient --no-cpu-baseline > $O/bench_blocked_rg.json 2> $O/bench_blocked_rg.err && cut -c1-300 $O/bench_blocked_rg.json
