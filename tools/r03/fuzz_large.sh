O=gpurun_out/r03; mkdir -p $O
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 330 --seed 411 --max-blocks 160 > $O/fuzz_large.log 2>&1; tail -2 $O/fuzz_large.log
