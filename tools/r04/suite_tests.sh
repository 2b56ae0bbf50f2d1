# round 4 final: the whole GPU test suite + fuzz.  usage: bash tools/r04/suite_tests.sh
set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/full_gpu.log 2>&1; tail -3 $O/full_gpu.log
timeout -k 10 420 python tools/fuzz_gpu.py --seconds 360 --seed 104 > $O/fuzz_gpu_360s.log 2>&1; tail -1 $O/fuzz_gpu_360s.log
bash tools/r03/fuzz_large.sh > $O/fuzz_large.log 2>&1; tail -2 $O/fuzz_large.log
