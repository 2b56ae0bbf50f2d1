set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests/test_gpu_stream.py tests/test_cli.py -m gpu -q -x > $O/stream_tests.log 2>&1; tail -2 $O/stream_tests.log
for i in 1 2; do python bench.py --stream --steps 30 --no-cpu-baseline --no-host-rate > $O/stream_bench$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/stream_bench$i.json')); print(d['roofline']['kernels_ms'], d['roofline']['frac'], d['config']['roundtrip_equals_pDecoded'])"; done
