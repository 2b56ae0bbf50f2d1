#!/bin/bash
# merged-block encoder: the merge's two rectangle passes on two host threads vs one (same box)
set -e
out=gpurun_out/r23; mkdir -p $out
nproc > $out/nproc.txt; lscpu | grep -i "model name\|thread\|core\|socket" >> $out/nproc.txt
python -m pytest tests/test_gpu_blocked.py -q -x -m gpu > $out/tests.txt 2>&1 || { tail -20 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
for mt in 1 0 1 0; do
  LIMG_MERGE_DEBUG=1 python bench.py --blocked --steps 8 --warmup 2 --merge-threads $mt --no-cpu-baseline > $out/b_mt$mt.json 2> $out/b_mt$mt.err || { tail $out/b_mt$mt.err; exit 1; }
  python - $out/b_mt$mt.json <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(l["config"]["merge_threads"], l["ms_per_step"], l["config"]["stage_ms"])
PY
  grep "setup+large" $out/b_mt$mt.err | tail -2
done
for mt in 1 0; do
  python bench.py --blocked --steps 8 --warmup 2 --merge-threads $mt --contexts 4 --no-cpu-baseline > $out/c4_mt$mt.json 2> $out/c4_mt$mt.err
  python - $out/c4_mt$mt.json <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(l["config"]["merge_threads"], l["ms_per_step"], l["config"]["pipelined_stream"])
PY
done
