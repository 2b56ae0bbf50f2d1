"""print the essentials of bench.py JSON lines: python tools/r06/show.py FILE..."""
import json
import sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f, d["value"], d["unit"], d["ms_per_step"], "ms", d["roofline"].get("frac"), d["roofline"].get("kernels_ms"), "errors", d["errors"])
            ow = d["config"].get("other_workloads")
            if ow:
                print("  headline verified:", d["config"].get("verified"))
                for k, v in ow.items():
                    print("  ", k, json.dumps(v)[:400])
