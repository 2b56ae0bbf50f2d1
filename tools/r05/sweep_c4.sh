O=gpurun_out/r05_c4sweep; mkdir -p $O
Q="--no-cpu-baseline --no-host-rate --config 4 --steps 3"
for sub in 4 8 16 -1; do python bench.py $Q --sub-images $sub > $O/sub$sub.json 2>/dev/null; done
for k in 0x50 0x60 0x40; do python bench.py $Q --pipeline-knobs $k > $O/knob$k.json 2>/dev/null; done
for k in 0x051 0x052; do python bench.py $Q --pipeline-knobs $k > $O/knob$k.json 2>/dev/null; done
python bench.py $Q --contexts 2 > $O/ctx2.json 2>/dev/null
python bench.py $Q --contexts 3 > $O/ctx3.json 2>/dev/null
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"].get("frac"), d.get("errors"))
    except Exception as e: print(os.path.basename(f), "UNREADABLE", e)
PY
