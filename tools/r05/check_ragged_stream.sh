mkdir -p gpurun_out/r05c; python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_gpu_rehearsal.py -x -q -m gpu > gpurun_out/r05c/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05c/tests.log; python tools/fuzz_gpu.py --seconds 150 > gpurun_out/r05c/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r05c/fuzz.log; Q="--no-cpu-baseline --no-host-rate"; python bench.py --stream $Q > gpurun_out/r05c/stream.json 2>gpurun_out/r05c/stream.err; for b in 0 -1 4 32; do python bench.py --steps 5 --warmup 1 --size 8190x8192 --ragged-bands $b $Q > gpurun_out/r05c/r8190_bands$b.json 2>/dev/null; done; python bench.py --steps 5 --warmup 1 --size 8190x8192 --pool-threads 2 $Q > gpurun_out/r05c/r8190_pool2.json 2>/dev/null;  python bench.py --steps 5 --warmup 1 --size 8190x8192 --pool-threads 2 --walk-threads 1 $Q > gpurun_out/r05c/r8190_pool2_serial.json 2>/dev/null; python bench.py --steps 5 --warmup 1 --size 8190x8192 --contexts 4 $Q > gpurun_out/r05c/r8190_ctx4.json 2>gpurun_out/r05c/r8190_ctx4.err; python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c/*.json")):
    try:
        l=json.load(open(f)); print(f, l["value"], l["ms_per_step"], l["roofline"]["kernels_ms"], l["config"].get("multi_context"), l["errors"])
    except Exception as e: print(f, "unreadable", e)
PY
