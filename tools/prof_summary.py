#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into a small text summary (kept under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-70s calls=%-5s avg_ns=%-12s total_ns=%-14s pct=%s" % (row.get("Name", "")[:70], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== PMC (mean per dispatch)")
for k, d in acc.items():
    print(" ", k)
    for c, v in sorted(d.items()):
        print("     %-28s %.4g   (n=%d)" % (c, sum(v) / len(v), len(v)))


# ---- counter-derived figures for bench.py, keyed by workload (profiles/pmc_by_workload.json) ----------------------------------------
# usage: prof_summary.py <dir> [--update-json profiles/pmc_by_workload.json [--kernel k_encode_persistent] [--source NAME]]
# The workload key is the one bench.py printed in the traced run's JSON line (roofline.pmc_key).
if "--update-json" in sys.argv:
    import json
    jpath = sys.argv[sys.argv.index("--update-json") + 1]
    want = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "k_encode_persistent"
    source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else os.path.basename(os.path.normpath(out))
    key = None
    try:
        for line in open(os.path.join(out, "trace.log")):
            line = line.strip()
            if line.startswith("{") and '"metric"' in line:
                key = json.loads(line)["roofline"].get("pmc_key")
    except Exception:
        pass
    m = None
    per_kernel = {}
    for w in want.split(","):  # several kernels make up one encode: their per-dispatch means add up
        for k, d in acc.items():
            if w in k:
                one = {c: sum(v) / len(v) for c, v in d.items()}
                per_kernel[w] = {"valu_instr": one.get("SQ_INSTS_VALU"), "salu_instr": one.get("SQ_INSTS_SALU"), "lds_instr": one.get("SQ_INSTS_LDS"),
                                 "fetch_kib": one.get("FETCH_SIZE"), "write_kib": one.get("WRITE_SIZE")}
                if m is None:
                    m = dict(one)
                else:
                    for c, v in one.items():
                        m[c] = m.get(c, 0.0) + v
    if key and m:
        entry = {"source": source, "kernel": want, "per_kernel": per_kernel,
                 "fetch_kib": m.get("FETCH_SIZE"), "write_kib": m.get("WRITE_SIZE"), "valu_instr_per_launch": m.get("SQ_INSTS_VALU"),
                 "salu_instr_per_launch": m.get("SQ_INSTS_SALU"), "lds_instr_per_launch": m.get("SQ_INSTS_LDS")}
        if m.get("SQ_ACTIVE_INST_VALU") and m.get("GRBM_GUI_ACTIVE"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            entry["valu_busy"] = round(m["SQ_ACTIVE_INST_VALU"] * 4.0 / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        if m.get("SQ_WAIT_INST_ANY") and m.get("SQ_WAVE_CYCLES"):
            entry["wait_inst_any_frac"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4)
        if m.get("SQ_LDS_BANK_CONFLICT") and m.get("SQ_LDS_IDX_ACTIVE"):
            entry["lds_bank_conflict_frac"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        try:
            allv = json.load(open(jpath))
        except Exception:
            allv = {}
        allv[key] = entry
        json.dump(allv, open(jpath, "w"), indent=1, sort_keys=True)
        print("== updated %s[%s]" % (jpath, key))
    else:
        print("== no json update: key=%r kernel found=%s" % (key, m is not None))
