O=gpurun_out/r03; mkdir -p $O
for rep in 1 2; do
for V in base prio0 prio3 nostagger; do
  LIMG_HIP_LIB=ab/$V/liblimg_hip.so python bench.py --steps 40 --no-cpu-baseline --no-host-rate > /tmp/ab_a.json 2>/dev/null
  LIMG_HIP_LIB=ab/$V/liblimg_hip.so python bench.py --size 4096 --workload random_gradient --steps 40 --no-cpu-baseline --no-host-rate > /tmp/ab_b.json 2>/dev/null
  LIMG_HIP_LIB=ab/$V/liblimg_hip.so python bench.py --config 4 --steps 3 --no-cpu-baseline --no-host-rate > /tmp/ab_c.json 2>/dev/null
  python - "$V" <<'PY'
import json, sys
a = json.load(open('/tmp/ab_a.json')); b = json.load(open('/tmp/ab_b.json')); c = json.load(open('/tmp/ab_c.json'))
print(sys.argv[1], a["ms_per_step"], a["roofline"]["kernels_ms"], "rg4096", b["ms_per_step"], "c4", c["ms_per_step"])
PY
done; done 2>&1 | tee $O/ab_misc.log
for W in 5 6; do LIMG_HIP_WG_PER_CU=$W python bench.py --steps 40 --no-cpu-baseline --no-host-rate 2>/dev/null | python -c "
import json,sys; a=json.loads(sys.stdin.read()); print('wg_per_cu', $W, a['ms_per_step'], a['roofline']['kernels_ms'])"; done 2>&1 | tee -a $O/ab_misc.log
