set -o pipefail
O=gpurun_out/r04_16; mkdir -p $O
B="--no-cpu-baseline --no-host-rate"
for sz in "1024x616 --rgb" "2048" "4096 --workload random_gradient" "8192"; do
  set -- $sz; tag=$1
  for rep in 1 2; do
  python bench.py $B --steps 200 --warmup 20 --size "$@" > $O/eager_${tag}_$rep.json 2>/dev/null
  python bench.py $B --steps 200 --warmup 20 --graph --size "$@" > $O/graph_${tag}_$rep.json 2>$O/graph_${tag}.err
  done
done
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("kernels_ms"))
    except Exception as e: print(os.path.basename(f), "failed", e)
PY
