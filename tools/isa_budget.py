#!/usr/bin/env python3
"""Per-phase instruction budget of the encode kernels, from the compiler's own assembly (hipcc -S -gline-tables-only) of the sources as they stand.

Every instruction of a kernel is attributed to a PHASE through its inlined-at chain (the .loc comments): the innermost frame that lies inside one of the phase
functions below decides.  A phase's static count is divided by the number of inlined copies the kernel holds of it (the search loop exists twice -- whole blocks and
blocks with masked lanes --, the F step's per-factor body six times ...), which gives the instructions of ONE execution; the last column multiplies by how often a
block executes the phase.  The dynamic totals (rocprofv3 SQ_INSTS_VALU / _SALU / _LDS per kernel, split path: one kernel per step) are printed next to the sums.

usage: python tools/isa_budget.py [--pmc gpurun_out/profiles/pmc_by_workload.json] > profiles/archive/r04_isa_budget.md"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "limg_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-gline-tables-only", "-S", "--cuda-device-only"]


def function_ranges(path):
    """name -> (first line, last line) of the top-level device functions / lambdas we care about, by brace matching from their first line."""
    src = open(path).read().split("\n")
    out = {}
    skip = {"__launch_bounds__", "__attribute__", "aligned", "noinline", "tpb_waves", "address_space"}
    i = 0
    while i < len(src):
        line = src[i]
        name = None
        if re.match(r"^\s*(?:template <.*>\s*)?(?:__device__|__global__)", line) and not line.rstrip().endswith(";"):
            for cand in re.findall(r"\b([A-Za-z_][A-Za-z_0-9]*)(?:<[^()]*>)?\(", line):
                if cand not in skip:
                    name = cand
                    break
        if name:
            j = i
            while "{" not in src[j]:
                j += 1
            if j == i and src[i].rstrip().endswith("}") and src[i].count("{") == src[i].count("}"):
                out.setdefault(name, (i + 1, i + 1))  # one-line function
                i += 1
                continue
            depth = 0
            k = j
            while True:
                depth += src[k].count("{") - src[k].count("}")
                if depth <= 0:
                    break
                k += 1
            out.setdefault(name, (i + 1, k + 1))
            i = k + 1
        else:
            i += 1
    return out


def marker(path, text, after=0):
    for n, line in enumerate(open(path), 1):
        if n > after and text in line:
            return n
    raise SystemExit("marker not found: " + text)


def assemble(src, extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-o", out, os.path.join(CSRC, src)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def kernel_instructions(asm, kernel_substr):
    """[(opcode, [(file, line) frames innermost first])] of one kernel"""
    on = False
    frames = []
    out = []
    for line in open(asm):
        if re.match(r"^_Z.*:", line):
            on = kernel_substr in line
            continue
        if not on:
            continue
        if ".end_amdhsa_kernel" in line or line.startswith(".Lfunc_end"):
            on = False
            continue
        if re.match(r"\s*\.loc\s", line):
            frames = [(os.path.basename(f), int(l)) for f, l in re.findall(r"([A-Za-z_0-9./]+\.(?:hip|h)):(\d+):\d+", line)]
            continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
        if not m or line.strip().startswith((".", ";")):
            continue
        out.append((m.group(1), frames))
    return out


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def budget(instrs, phases, default):
    acc = collections.OrderedDict((name, collections.Counter()) for name, _ in phases)
    acc[default] = collections.Counter()
    ops = collections.defaultdict(collections.Counter)
    for op, frames in instrs:
        hit = default
        for f, l in frames:
            for name, ranges in phases:
                if any(ff == f and a <= l <= b for ff, a, b in ranges):
                    hit = name
                    break
            if hit != default:
                break
        acc[hit][kind(op)] += 1
        ops[hit][op] += 1
    return acc, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "pmc_by_workload.json"))
    ap.add_argument("--calibration", default=os.path.join(ROOT, "profiles", "r04_isa_calibration.json"), help="measured aggregates from tools/isa_calibrate.py")
    ap.add_argument("--trials", type=float, default=12.1, help="trials per block of the workload (oracle statistics: 8192^2 photo-noise, errorFactor 100)")
    ap.add_argument("--rebuilds", type=float, default=19.4, help="factor rebuilds per block that are not to shift 8 (same statistics; the first triple's three included in 'trial set-up')")
    ap.add_argument("--sums", type=float, default=10.0, help="block-error sums per block (trials that no pixel fails)")
    ap.add_argument("--dithers", type=float, default=2.2, help="dithered factors per block (shifts 1..7)")
    args = ap.parse_args()
    K = os.path.join(CSRC, "limg_hip_kernels.hip")
    D = os.path.join(CSRC, "limg_hip_device.h")
    fk, fd = function_ranges(K), function_ranges(D)
    kf = "limg_hip_kernels.hip"

    def fn(name, table=fk, file=kf):
        a, b = table[name]
        return (file, a, b)

    # sub-ranges of fit_search_strip by markers in the source
    e0, e1 = fk["fit_search_strip"]
    m_stage = marker(K, "// ---- stage: the strip's pixel rows into LDS")
    m_prefit = marker(K, "the records of the wave's 8 blocks as k_fit_tpb left them")
    m_float = marker(K, "// The float stage runs in batches of kBatch blocks per wave")
    m_view = marker(K, "// phase-E view (overlays the dead float-stage fields)")
    m_phaseE = marker(K, "// ---- phase E: per-pixel factors (a8) + shift search (a10-a12)")
    m_a8 = marker(K, "{ // a8 (src/limg_factorization.h:149-197)")
    m_shift = marker(K, "uint32_t shift[3] = { 0, 0, 0 };", after=m_a8)
    m_calls = marker(K, "// dither calls this block will make (src/limg.cpp:1951-1958)")
    m_after = marker(K, "if (lane == 0) s_calls[wave] = waveCalls;")
    asm = assemble("limg_hip_kernels.hip", ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"])
    asm_fit = assemble("limg_hip_fit_tpb.hip")
    print("# Per-phase instruction budget of the encode path (generated by tools/isa_budget.py from the assembly of the committed sources)\n")
    print("Kernel variants of the headline workload: `k_fit_tpb<4, false, true>`, `k_encode_persistent<4, false, true, false>` (RGBA, EXACT float stage, records from k_fit_tpb, default search).")
    print("Static counts are instructions in the `.s`; 'one execution' divides by the inlined copies of the phase; 'per block' multiplies by executions per 8x8 block")
    print("(workload statistics from the oracle: %.1f trials, %.1f real factor rebuilds, %.1f block sums, %.1f dithered factors per block).\n" % (args.trials, args.rebuilds, args.sums, args.dithers))

    phases = [
        ("search: trial core (a9)", [fn("trial_pixel_error")]),
        ("search: factor rebuild (make_terms)", [fn("make_terms"), fn("rebuild_A"), fn("rebuild_B"), fn("rebuild_C")]),
        ("search: block-error sum (wave_sum)", [("limg_hip_device.h",) + fd["wave_sum"]]),
        ("search: automaton loop / entry load", [fn("search_fast_automaton"), fn("sload8")]),
        ("search: accurate automaton (other variant)", [fn("search_accurate_automaton")]),
        ("E: strip staging (pixels -> LDS)", [(kf, m_stage, m_prefit - 1)]),
        ("E: record load + flags (PREFIT)", [(kf, m_prefit, m_float - 1)]),
        ("E: lane == pixel float stage (other variant)", [(kf, m_float, m_view - 1), fn("serial_sums2")]),
        ("E: phase-E view + trial constants (a7)", [(kf, m_view, m_phaseE - 1)]),
        ("E: per-pixel factors (a8)", [(kf, m_a8, m_shift - 1)]),
        ("E: block loop, trial set-up", [(kf, m_phaseE, m_a8 - 1), (kf, m_shift, m_calls - 1)]),
        ("E: block epilogue (shift word, factor bytes -> LDS)", [(kf, m_calls, m_after - 1)]),
        ("E: strip epilogue (descriptor, park)", [(kf, m_after, e1)]),
        ("F: prepare (plane constants, decode constants, flags)", [fn("phase_f_prepare")]),
        ("F: 7 uniform planes' stores", [fn("phase_f_store_const")]),
        ("F: chain position (look-back, first calls)", [fn("lookback_base"), fn("phase_f_first_calls"), fn("desc_load"), fn("desc_store")]),
        ("F: dither + crushed bytes + decode terms (rows_factor)", [fn("rows_factor"), fn("add_byte_sdwa"), fn("and_into_byte_sdwa")]),
        ("F: row set-up, clamp + pack, stores (phase_f_rows)", [fn("phase_f_rows")]),
        ("F: lane == pixel pixels phase (ragged images only)", [fn("phase_f_pixels")]),
        ("F: strip load (park -> LDS), rest of dither_store_strip", [fn("dither_store_strip")]),
        ("persistent loop (ticket, barriers)", [fn("k_encode_persistent")]),
    ]
    instrs = kernel_instructions(asm, "k_encode_persistentILi4ELb0ELb1ELb0E")
    acc, ops = budget(instrs, phases, "unattributed")
    copies_trial = max(1, ops["search: trial core (a9)"]["v_dot2_u32_u16"])
    copies_rows = max(1, ops["F: dither + crushed bytes + decode terms (rows_factor)"]["v_add_u32_sdwa"] // 8)
    copies_sum = max(1, ops["search: block-error sum (wave_sum)"]["v_readlane_b32"])
    copies_rebuild = max(1, ops["search: factor rebuild (make_terms)"]["v_perm_b32"])
    # (name) -> (copies, executions per block, note)
    T, R, S, Dz = args.trials, args.rebuilds, args.sums, args.dithers
    how = {
        "search: trial core (a9)": (copies_trial, T, "per trial"),
        "search: factor rebuild (make_terms)": (copies_rebuild, R + 3, "per rebuilt factor (+ the first triple's three)"),
        "search: block-error sum (wave_sum)": (copies_sum, S + 1.0 / 32, "per trial no pixel fails (+ the F step's one per strip)"),
        "search: automaton loop / entry load": (2, T, "per trial (two copies: whole blocks / masked lanes)"),
        "search: accurate automaton (other variant)": (1, 0, "not in this variant"),
        "E: strip staging (pixels -> LDS)": (1, 1.0 / 8, "per wave and strip (8 blocks)"),
        "E: record load + flags (PREFIT)": (1, 1.0 / 8, "per wave and strip"),
        "E: lane == pixel float stage (other variant)": (1, 0, "not in this variant"),
        "E: phase-E view + trial constants (a7)": (1, 1.0 / 8, "per wave and strip"),
        "E: per-pixel factors (a8)": (1, 1, "per block"),
        "E: block loop, trial set-up": (1, 1, "per block (both search copies' set-up counted once each)"),
        "E: block epilogue (shift word, factor bytes -> LDS)": (1, 1, "per block"),
        "E: strip epilogue (descriptor, park)": (1, 1.0 / 8, "per wave and strip"),
        "F: prepare (plane constants, decode constants, flags)": (1, 1.0 / 8, "per wave and strip"),
        "F: 7 uniform planes' stores": (1, 1.0 / 8, "per wave and strip"),
        "F: chain position (look-back, first calls)": (1, 1.0 / 32, "wave 0, per strip"),
        "F: dither + crushed bytes + decode terms (rows_factor)": (copies_rows, 3.0 / 8, "per factor and wave pass (8 blocks x 8 px per lane)"),
        "F: row set-up, clamp + pack, stores (phase_f_rows)": (1, 1.0 / 8, "per wave and strip (alpha / generic branches counted in)"),
        "F: lane == pixel pixels phase (ragged images only)": (1, 0, "not executed for whole blocks"),
        "F: strip load (park -> LDS), rest of dither_store_strip": (1, 1.0 / 8, "per wave and strip"),
        "persistent loop (ticket, barriers)": (1, 1.0 / 8, "per wave and strip"),
        "unattributed": (1, 1.0 / 8, "compiler-generated (address arithmetic, spills): per wave and strip assumed"),
    }
    print("## k_encode_persistent<4, false, true, false>\n")
    print("| phase | static VALU / SALU / LDS / VMEM | copies | one execution: VALU / SALU / LDS | executions per block | per block: VALU / SALU / LDS |")
    print("|---|---|---|---|---|---|")
    tot = collections.Counter()
    for name, c in acc.items():
        copies, per_block, note = how[name]
        one = {k: c[k] / copies for k in ("valu", "salu", "lds")}
        pb = {k: one[k] * per_block for k in one}
        for k in pb:
            tot[k] += pb[k]
        print("| %s | %d / %d / %d / %d | %d | %.1f / %.1f / %.1f | %.3g (%s) | %.1f / %.1f / %.1f |" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], copies, one["valu"], one["salu"], one["lds"],
                                                                                              per_block, note, pb["valu"], pb["salu"], pb["lds"]))
    print("| **sum** | | | | | **%.0f / %.0f / %.0f** |" % (tot["valu"], tot["salu"], tot["lds"]))
    # ---- reconciliation with MEASURED aggregates (tools/isa_calibrate.py: errorFactor sweep + search bypassed) ----
    print("\nThe static columns count every instruction of a phase's code, whichever way its branches go.  For straight-line vector code that is what executes; for scalar code it")
    print("is an UPPER BOUND -- guards (`if (e[0] & 0x20) rebuild_A`), both tails of the trial, both copies of the search loop, the generic-path and alpha branches of the F step")
    print("are all in the count although a block runs one side of each.  The scalar column is therefore not a budget; what can be checked is the two aggregates the counters")
    print("measure directly (`profiles/archive/r04_isa_calibration.md`): the block with the search bypassed, and the all-in cost of a trial.\n")
    try:
        cal = json.load(open(args.calibration))
        search_rows = [n for n in acc if n.startswith("search:")]
        setup_row = "E: block loop, trial set-up"  # (skipped when the search is bypassed: it belongs to the search's per-block cost)
        st_search = collections.Counter()
        st_setup = collections.Counter()
        st_rest = collections.Counter()
        for name, c in acc.items():
            copies, per_block, _ = how[name]
            for k in ("valu", "salu", "lds"):
                (st_search if name in search_rows else (st_setup if name == setup_row else st_rest))[k] += c[k] / copies * per_block
        m0, ms, mt = cal["bypassed_per_block"], cal["search_setup_per_block"], cal["per_trial_slope"]
        print("| aggregate | static VALU / SALU / LDS | measured VALU / SALU / LDS | static / measured |")
        print("|---|---|---|---|")
        print("| everything but the search, per block (measured: `--forced-shift 0`) | %.0f / %.0f / %.0f | %.0f / %.0f / %.0f | %.2f / %.2f / %.2f |"
              % (st_rest["valu"], st_rest["salu"], st_rest["lds"], m0[0], m0[1], m0[2], st_rest["valu"] / m0[0], st_rest["salu"] / m0[1], st_rest["lds"] / m0[2]))
        print("| search set-up, per block (measured: intercept of the errorFactor sweep) | %.0f / %.0f / %.0f | %.0f / %.0f / %.0f | %.2f / %.2f / %.2f |"
              % (st_setup["valu"], st_setup["salu"], st_setup["lds"], ms[0], ms[1], ms[2], st_setup["valu"] / ms[0], st_setup["salu"] / ms[1], st_setup["lds"] / max(ms[2], 1e-9)))
        print("| the search, per trial (measured: slope of the sweep; %.1f trials per block) | %.1f / %.1f / %.2f | %.1f / %.1f / %.2f | %.2f / %.2f / - |"
              % (T, st_search["valu"] / T, st_search["salu"] / T, st_search["lds"] / T, mt[0], mt[1], mt[2], st_search["valu"] / T / mt[0], st_search["salu"] / T / mt[1]))
        print("| whole kernel, per block | %.0f / %.0f / %.0f | %.0f / %.0f / %.0f | %.2f / %.2f / %.2f |"
              % (tot["valu"], tot["salu"], tot["lds"], m0[0] + ms[0] + T * mt[0], m0[1] + ms[1] + T * mt[1], m0[2] + ms[2] + T * mt[2],
                 tot["valu"] / (m0[0] + ms[0] + T * mt[0]), tot["salu"] / (m0[1] + ms[1] + T * mt[1]), tot["lds"] / (m0[2] + ms[2] + T * mt[2])))
        print("\nVector side: the static trial (core + %.2f rebuilds + %.2f sums) is within %.0f %% of the measured slope; the non-search code over-counts by the branches a block does not"
              % (R / T, S / T, 100 * abs(st_search["valu"] / T / mt[0] - 1)))
        print("take (the F step's generic / alpha paths, the second search copy's set-up).  Scalar side: a trial executes %.1f scalar instructions, 1 / %.1f of what its code holds." % (mt[1], st_search["salu"] / T / mt[1]))
    except Exception as e:
        print("(no calibration file: %r)" % (e,))
    try:
        pmc = json.load(open(args.pmc))
        fused = pmc["8192x8192_photo_noise_ef100_fused"]["per_kernel"]
        split = pmc.get("8192x8192_photo_noise_ef100_split", {}).get("per_kernel", {})
        B = 1048576.0
        print("\nMeasured (rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS, 8192^2 photo-noise, per 8x8 block; source `%s`):\n" % pmc["8192x8192_photo_noise_ef100_fused"].get("source"))
        print("| kernel | VALU | SALU | LDS |\n|---|---|---|---|")
        for k, v in list(fused.items()) + [("(split path) " + k, v) for k, v in split.items()]:
            print("| %s | %.1f | %.1f | %.1f |" % (k, v["valu_instr"] / B, v["salu_instr"] / B, v["lds_instr"] / B))
        print("\nThe split path runs the E step (`k_fit_search`) and the F step (`k_dither_store`) as kernels of their own: their counts are the measured size of the two halves of the table above.")
    except Exception as e:
        print("\n(no PMC file: %r)" % (e,))

    # ---- k_fit_tpb
    F = os.path.join(CSRC, "limg_hip_fit_tpb.hip")
    ff = "limg_hip_fit_tpb.hip"
    p1 = marker(F, "// ---- pass 1 (src/limg_factorization.h:602-628)")
    p2 = marker(F, "// ---- pass 2 (:652-688)")
    p3 = marker(F, "// ---- pass 3 ----")
    p4 = marker(F, "// ---- pass 4 (:748-758) ----")
    p3c = marker(F, "{ // 3 channels (:498-541)")
    prec = marker(F, "// ---- record (src/limg_factorization.h:764-790)")
    psum = marker(F, "// ---- a4: channel sums (src/limg.cpp:466-497)")
    fend = function_ranges(F)["k_fit_tpb"][1]
    phases = [
        ("a4 channel sums", [(ff, psum, p1 - 1)]),
        ("pass 1 (unit vectors of px - avg)", [(ff, p1, p2 - 1)]),
        ("pass 2 (factor A extrema, residual direction)", [(ff, p2, p3 - 1)]),
        ("pass 3 (factor B extrema, residual direction; 4 ch)", [(ff, p3, p4 - 1)]),
        ("pass 4 (factor C extrema; 4 ch)", [(ff, p4, p3c - 1)]),
        ("3-channel pass 3 (other variant)", [(ff, p3c, prec - 1)]),
        ("record", [(ff, prec, fend)]),
    ]
    instrs = kernel_instructions(asm_fit, "k_fit_tpbILi4ELb0ELb1E")
    acc, ops = budget(instrs, phases, "prologue (table -> LDS, addressing)")
    print("\n## k_fit_tpb<4, false, true>  (one lane per block: a wave instruction counts 1/64 per block)\n")
    print("The four passes are loops over the block's 8 rows (`#pragma unroll 1`, 8 pixels unrolled inside): static count x 8 iterations / 64 blocks per wave = static / 8 per block.\n")
    print("| phase | static VALU / SALU / LDS / VMEM | per block VALU |")
    print("|---|---|---|")
    tv = 0.0
    for name, c in acc.items():
        loops = name.startswith("pass") or name.startswith("a4")
        pb = c["valu"] * (8 if name.startswith("pass") else 1) / 64.0
        tv += pb
        print("| %s | %d / %d / %d / %d | %.1f%s |" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], pb, " (loop body x 8)" if name.startswith("pass") else ""))
    print("| **sum** | | **%.0f** |" % tv)
    for f in (asm, asm_fit):
        os.unlink(f)


if __name__ == "__main__":
    main()
