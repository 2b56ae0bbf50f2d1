#!/usr/bin/env python3
"""Calibrate the per-execution instruction costs of k_encode_persistent's search loop from MEASURED dynamic counts (the static per-phase
budget of tools/isa_budget.py cannot weigh branch targets by execution, so its scalar column summed to 3.5 x the counters).

Input: rocprofv3 --pmc SQ_INSTS_VALU / _SALU / _LDS of the default bench on workloads whose search statistics differ (tools/r04/pmc_calib.sh: errorFactor 25 ... 400,
search bypassed with forced shifts 0 and 4) + the per-block statistics of those workloads from the oracle (tools/search_stats.py).  Model, per 8x8 block:
    count = c0 + cT * trials + cR * real factor rebuilds + c8 * rebuilds to shift 8 + cS * block-error sums + cD * dithered factors
Least squares over the workloads, for each of VALU / SALU / LDS; prints the coefficients, the per-workload residuals, and the headline workload's break-down.
usage: python tools/isa_calibrate.py gpurun_out/r04_calib profiles/archive/r04_search_stats.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import numpy as np


def counts(d, kernel="k_encode_persistent"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    out, stats_path = sys.argv[1], sys.argv[2]
    stats = json.load(open(stats_path))["per_block"]
    blocks = 1024.0 * 1024.0
    rows, meas, names = [], [], []
    for tag in ("ef25", "ef50", "ef100", "ef200", "ef400", "shift0", "shift4"):
        c = counts(os.path.join(out, tag))
        if not c:
            continue
        if tag.startswith("ef"):
            s = stats[tag]
            x = [1.0, s["trials"], s["rebuild_real"], s["rebuild_to8"], s["sums"], s["dither_calls"]]
        else:
            x = [1.0, 0.0, 0.0, 0.0, 0.0, 3.0 if tag == "shift4" else 0.0]
        rows.append(x); names.append(tag)
        meas.append([c.get("SQ_INSTS_VALU", 0) / blocks, c.get("SQ_INSTS_SALU", 0) / blocks, c.get("SQ_INSTS_LDS", 0) / blocks])
    if len(rows) < 6:
        print("(not enough workloads measured: %r)" % names)
        return
    Y = np.array(meas)
    # The statistics of the errorFactor sweep move together (more trials = more rebuilds = more sums), so separate costs per rebuild / per sum are not identifiable
    # from it (a six-parameter least-squares fit reproduces every workload to 0.3 % with meaningless, partly negative coefficients -- tried).  What the sweep does
    # pin down, and what the static budget has to be checked against, are two aggregates: the block's cost with the search bypassed, and the all-in cost of a trial.
    i0 = names.index("shift0")
    c0 = Y[i0]
    print("## Measured aggregates of `k_encode_persistent<4, false, true, false>` (rocprofv3 --pmc SQ_INSTS_VALU / _SALU / _LDS; instructions per 8x8 block)\n")
    print("| workload | trials / real rebuilds / to-8 / sums / dithers per block | VALU / SALU / LDS per block | minus the search-bypassed block | per trial, all-in (VALU / SALU / LDS) |\n|---|---|---|---|---|")
    per_trial = {}
    for n, x, y in zip(names, rows, Y):
        d = y - c0
        if x[1] > 0:
            per_trial[n] = d / x[1]
            pt = "%.1f / %.1f / %.2f" % tuple(per_trial[n])
        else:
            pt = "-"
        print("| %s | %.1f / %.1f / %.1f / %.1f / %.1f | %.1f / %.1f / %.1f | %.1f / %.1f / %.1f | %s |" % (n, x[1], x[2], x[3], x[4], x[5], y[0], y[1], y[2], d[0], d[1], d[2], pt))
    print("\nSearch bypassed (`--forced-shift 0`): **%.0f VALU + %.0f SALU + %.0f LDS per block** -- staging, record load, a7 view, a8, block set-up / epilogue, the whole F step." % tuple(c0))
    if "ef100" in per_trial:
        print("A trial of the headline workload, everything included (core, %.2f real rebuilds, %.2f block sums, loop control): **%.1f VALU + %.1f SALU + %.2f LDS**."
              % (rows[names.index("ef100")][2] / rows[names.index("ef100")][1], rows[names.index("ef100")][4] / rows[names.index("ef100")][1], *per_trial["ef100"]))
    # two-parameter fit over the errorFactor sweep: (count - bypassed block) = search set-up per block + slope * trials  (well conditioned, unlike the six-parameter one)
    efs = [i for i, n in enumerate(names) if n.startswith("ef")]
    A = np.array([[1.0, rows[i][1]] for i in efs])
    D = np.array([Y[i] - c0 for i in efs])
    fit2, *_ = np.linalg.lstsq(A, D, rcond=None)
    res = A @ fit2 - D
    print("Linear in the trial count over the sweep: search set-up **%.1f VALU + %.1f SALU + %.1f LDS per block**, then **%.1f VALU + %.1f SALU per trial** (largest residual %.1f VALU / %.1f SALU per block)."
          % (fit2[0][0], fit2[0][1], fit2[0][2], fit2[1][0], fit2[1][1], np.abs(res[:, 0]).max(), np.abs(res[:, 1]).max()))
    json.dump({"search_setup_per_block": fit2[0].tolist(), "per_trial_slope": fit2[1].tolist(), "bypassed_per_block": c0.tolist(), "per_trial": {k: v.tolist() for k, v in per_trial.items()}, "workloads": names, "x": rows, "measured": meas},
              open(os.path.join(out, "calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
