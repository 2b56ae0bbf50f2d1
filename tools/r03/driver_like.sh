set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
S=$(date +%s); python bench.py > $O/driver_bench.json 2> $O/driver_bench.err; echo "bench wall seconds: $(( $(date +%s) - S ))"; python -c "
import json; d=json.load(open('$O/driver_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['valu'] and d['roofline']['valu']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config']['cold'])"
