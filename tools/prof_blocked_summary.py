#!/usr/bin/env python3
"""Condenses a tools/prof_blocked.sh directory PER IMAGE: for every kernel of the merged-block encoder the dispatches, the GPU time (kernel trace) and the counter
totals of one image = sums over all dispatches of the run / images encoded in it.  usage: prof_blocked_summary.py <dir> <images> [--update-json FILE --source NAME]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, images = sys.argv[1], int(sys.argv[2])
KERNELS = ("k_fit_tpb", "k_fit_search", "k_blocked_bounds", "k_blocked_match", "k_blocked_fit_search", "k_noise_expand_calls", "k_blocked_store")


def short(name):
    for k in sorted(KERNELS, key=len, reverse=True):
        if k in name:
            return k
    return None


per = {k: defaultdict(float) for k in KERNELS}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row.get("Name", ""))
        if k:
            per[k]["dispatches"] += float(row["Calls"]) / images
            per[k]["ms"] += float(row["TotalDurationNs"]) / 1e6 / images
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k:
            per[k][row["Counter_Name"]] += float(row["Counter_Value"]) / images
line = None
try:
    for raw in open(os.path.join(out, "trace.log")):
        if raw.startswith("{") and '"metric"' in raw:
            line = json.loads(raw)
except Exception:
    pass
print("== merged-block encoder, per image (%d images in the run)" % images)
for k in KERNELS:
    d = per[k]
    if not d:
        continue
    print("  %-22s dispatches %-6.1f ms %-8.3f VALU %-10.4g SALU %-10.4g LDS %-10.4g waves %-9.4g fetch MiB %-8.1f write MiB %-8.1f wait_inst_any/wave_cycles %.3f valu_busy(own time) %s"
          % (k, d["dispatches"], d["ms"], d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"], d["SQ_WAVES"], 2 * d["FETCH_SIZE"] / 1024, d["WRITE_SIZE"] / 1024,
             (d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]) if d["SQ_WAVE_CYCLES"] else float("nan"),
             ("%.3f" % (d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))) if d["GRBM_GUI_ACTIVE"] else "n/a"))
if "--update-json" in sys.argv and line:
    jpath = sys.argv[sys.argv.index("--update-json") + 1]
    source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else os.path.basename(os.path.normpath(out))
    cfg = line["config"]["workload"]
    key = line["roofline"].get("pmc_key")
    entry = {"source": source, "per_image": True, "images_in_run": images,
             "per_kernel": {k: {"dispatches": round(d["dispatches"], 2), "ms": round(d["ms"], 4), "valu_instr": d["SQ_INSTS_VALU"], "salu_instr": d["SQ_INSTS_SALU"], "lds_instr": d["SQ_INSTS_LDS"],
                                "waves": d["SQ_WAVES"], "fetch_kib": d["FETCH_SIZE"], "write_kib": d["WRITE_SIZE"],
                                "wait_inst_any_frac": round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4) if d["SQ_WAVE_CYCLES"] else None,
                                "valu_busy": round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4) if d["GRBM_GUI_ACTIVE"] else None}
                            for k, d in per.items() if d},
             "fetch_kib": sum(d["FETCH_SIZE"] for d in per.values()), "write_kib": sum(d["WRITE_SIZE"] for d in per.values()),
             "valu_instr_per_launch": sum(d["SQ_INSTS_VALU"] for d in per.values())}
    try:
        allv = json.load(open(jpath))
    except Exception:
        allv = {}
    if key:
        allv[key] = entry
        json.dump(allv, open(jpath, "w"), indent=1, sort_keys=True)
        print("== updated %s[%s] (%s)" % (jpath, key, cfg))
