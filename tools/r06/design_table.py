#!/usr/bin/env python3
"""Prints DESIGN.md section 5's table from the bench lines and rocprof summaries under profiles/ (so that the document quotes the files, not memory).
usage: python tools/r06/design_table.py [--write]   (--write replaces the table between the markers in DESIGN.md)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles")


def line(name):
    with open(os.path.join(P, "r06_bench_final%s.json" % ("" if name == "default" else "_" + name))) as f:
        for raw in f:
            if raw.startswith("{"):
                return json.loads(raw)


def kstats(tag):
    out = {}
    for r in csv.DictReader(open(os.path.join(P, "%s_kernel_stats.csv" % tag))):
        for k in ("k_fit_tpb", "k_encode_persistent", "k_stream_scan_strips", "k_stream_pack_strips", "k_stream_decode"):
            if k in r["Name"]:
                out[k] = float(r["AverageNs"]) / 1e6
    return out


def main():
    d = line("default")
    rf, cfg = d["roofline"], d["config"]
    ks = kstats("r06_final")
    km = list(rf["kernels_ms"].values())
    cb = d["cpu_baseline"]["builds_Mpixels_per_s"]["fastmath"]
    rows = ["| workload | ms | Mpx/s | of HBM roofline | source |", "|---|---|---|---|---|"]
    rows.append("| **headline: 8192² RGBA photo-noise, ef 100, fast, one chain, 11 planes** | %.4f | **%s** | **%.4f** (kernels %.4f + %.4f; rocprofv3 %.4f + %.4f = %.3f) | `r06_bench_final.json`, `r06_final_kernel_stats.csv` |"
                % (d["ms_per_step"], "{:,.0f}".format(d["value"]).replace(",", " "), rf["frac"], km[0], km[1], ks["k_fit_tpb"], ks["k_encode_persistent"],
                   rf["algorithmic_bytes_per_launch"] / ((ks["k_fit_tpb"] + ks["k_encode_persistent"]) * 1e-3) / 8e12))
    rows.append("| — traffic / VALU | | | %.3f GB moved = %.2f × algorithmic; %.1f VALU / block at %.2f of the issue ceiling, busy %.2f; instruction floor %.4f | `r06_final_summary.txt`, `pmc_by_workload.json` |"
                % (rf["traffic"] / 1e9, rf["traffic"] / rf["algorithmic_bytes_per_launch"], rf["valu"]["instr_per_block"], rf["valu"]["frac"], rf["valu"]["valu_busy"], rf["instruction_floor"]))
    rows.append("| — CPU beside it | | %.1f (reference, its best pool: %d threads; %.1f at all host threads; %.1f single thread) | GPU = %.0f × | `r06_bench_final.json` `cpu_baseline` |"
                % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], cb["test_pool_allcores"], cb["test_1thread"], d["value"] / d["cpu_baseline"]["value"]))
    ow = cfg["other_workloads"]
    rows.append("| — on the same driver-run line, each verified against the reference (`config.other_workloads`) | | config 2 %.0f, config 4 %.0f, config 5 strip %.0f / %.0f, stream %.0f, merged-block %.0f | %.3f, %.3f, %.3f / %.3f | `r06_bench_final.json` |"
                % (ow["config2_rg4096"]["Mpixels_per_s"], ow["config4_batch64_rg4096"]["Mpixels_per_s"], ow["config5_strip_pool0"]["Mpixels_per_s"], ow["config5_strip_pool2"]["Mpixels_per_s"],
                   ow["stream_pn8192"]["Mpixels_per_s"], ow["blocked_pn8192"]["Mpixels_per_s"], ow["config2_rg4096"]["frac"], ow["config4_batch64_rg4096"]["frac"],
                   ow["config5_strip_pool0"]["frac"], ow["config5_strip_pool2"]["frac"]))

    def simple(label, names, src):
        ls = [line(n) for n in names]
        rows.append("| %s | %s | %s | %s | %s |" % (label, " / ".join("%.3f" % l["ms_per_step"] for l in ls), " / ".join("{:,.0f}".format(l["value"]).replace(",", " ") for l in ls),
                                                " / ".join("%.4f" % l["roofline"]["frac"] for l in ls), src))
    simple("config 2: 4096² gradient", ["rg4096"], "`r06_bench_final_rg4096.json`")
    simple("config 3 sweep ends: ef 25 / ef 400", ["ef25", "ef400"], "`r06_bench_final_ef25.json`, `_ef400.json`")
    simple("config 4: 64 × 4096² batch, one GPU", ["c4"], "`r06_bench_final_c4.json`")
    simple("config 5 on one GPU: 8 strips restarted / one chain", ["c5", "c5_single_chain"], "`r06_bench_final_c5.json`, `_c5_single_chain.json`")
    a = line("accurate")
    pa = json.load(open(os.path.join(P, "pmc_by_workload.json")))["8192x8192_photo_noise_ef100_fused_accurate"]
    rows.append("| accurate search 8192² / 4096² gradient | %.3f / %.3f | %s / %s | %.4f (%.0f VALU / block, busy %.2f) / %.4f | `r06_bench_final_accurate.json`, `_accurate_rg4096.json`, `r06_final_accurate_summary.txt` |"
                % (a["ms_per_step"], line("accurate_rg4096")["ms_per_step"], "{:,.0f}".format(a["value"]).replace(",", " "), "{:,.0f}".format(line("accurate_rg4096")["value"]).replace(",", " "),
                   a["roofline"]["frac"], pa["valu_instr_per_launch"] / 1048576, pa["valu_busy"], line("accurate_rg4096")["roofline"]["frac"]))
    simple("RGB (3 channels) / FAST float / split path / lane == pixel float stage", ["rgb", "fast", "split", "legacy"], "`r06_bench_final_rgb.json`, `_fast`, `_split`, `_legacy`")
    simple("8192 × 8190 / 8190 × 8192 / the same with a pool of 2 / 1024 × 618 RGB", ["8192x8190", "8190x8192", "8190x8192_pool2", "1024x618"], "`r06_bench_final_8192x8190.json` …")
    s = line("stream")
    st = kstats("r06_final_stream")
    rows.append("| stream: encode + pack / pack alone / decode | %.3f / %.4f / %.4f (rocprofv3: scan %.4f + pack %.4f = %.4f; decode %.4f) | %s | pack %.2f, decode %.2f (rocprofv3 %.2f) | `r06_bench_final_stream.json`, `r06_final_stream_kernel_stats.csv` |"
                % (s["ms_per_step"], s["roofline"]["kernels_ms"]["k_stream_count+scan+pack"], s["roofline"]["kernels_ms"]["k_stream_decode"], st["k_stream_scan_strips"], st["k_stream_pack_strips"],
                   st["k_stream_scan_strips"] + st["k_stream_pack_strips"], st["k_stream_decode"], "{:,.0f}".format(s["value"]).replace(",", " "),
                   5.9 * 8192 * 8192 / ((st["k_stream_scan_strips"] + st["k_stream_pack_strips"]) * 1e-3) / 8e12, s["roofline"]["frac"],
                   s["roofline"]["algorithmic_bytes_per_launch"] / (st["k_stream_decode"] * 1e-3) / 8e12))
    b, brg = line("blocked"), line("blocked_rg")
    rows.append("| merged-block encoder, photo-noise / gradient: one image; four contexts (8+ images each) | %.1f / %.1f | %s / %s; %s / %s | kernel-only %.3f / %.3f | `r06_bench_final_blocked.json`, `_blocked_rg.json` |"
                % (b["ms_per_step"], brg["ms_per_step"], "{:,.0f}".format(b["value"]).replace(",", " "), "{:,.0f}".format(brg["value"]).replace(",", " "),
                   "{:,.0f}".format(b["config"]["pipelined_stream"]["Mpixels_per_s"]).replace(",", " "), "{:,.0f}".format(brg["config"]["pipelined_stream"]["Mpixels_per_s"]).replace(",", " "),
                   b["roofline"]["frac"], brg["roofline"]["frac"]))
    he, hp = cfg["host_entry"], line("host_pool2")["config"]["host_entry"]
    rows.append("| host-pointer entry 8192² (PCIe-inclusive; never `value`), poolThreads 0 / 2 | %.2f / %.2f | %s / %s | — (%.1f GB/s over PCIe) | `r06_bench_final.json`, `_host_pool2.json` `config.host_entry` |"
                % (he["ms"], hp["ms"], "{:,.0f}".format(he["Mpixels_per_s"]).replace(",", " "), "{:,.0f}".format(hp["Mpixels_per_s"]).replace(",", " "), he["GB_per_s"]))
    if he.get("perf_entry"):
        pe = he["perf_entry"]
        rows.append("| `limg_hip_encode3d_perf` through host pointers (4 B/px up, nothing stored: the reference tool's `--count` loop) | %.2f | %s | — (%.1f GB/s over PCIe) | `r06_bench_final.json` `config.host_entry.perf_entry` |"
                    % (pe["ms"], "{:,.0f}".format(pe["Mpixels_per_s"]).replace(",", " "), pe["GB_per_s"]))
    rows.append("| two contexts on two streams, images alternating / first encode of a fresh context / `_perf` style (E step only) | %.3f per image / %.1f / %.3f | %s / — / %s | | `r06_bench_final.json` `config.two_streams`, `first_encode_ms`, `perf_style_ms` |"
                % (cfg["two_streams"]["ms_per_image"], cfg["first_encode_ms"], cfg["perf_style_ms"], "{:,.0f}".format(cfg["two_streams"]["Mpixels_per_s"]).replace(",", " "),
                   "{:,.0f}".format(cfg["perf_style_Mpixels_per_s"]).replace(",", " ")))
    cc = cfg.get("cold_context_per_image")
    if cc:
        rows.append("| a context per image: init + one encode + shutdown | %.2f | %s | (context: %.0f MB of device memory) | `r06_bench_final.json` `config.cold_context_per_image` |"
                    % (cc["ms_per_image"], "{:,.0f}".format(cc["Mpixels_per_s"]).replace(",", " "), cc["context_device_bytes"] / 1e6))
    text = "\n".join(rows)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a0, b0 = s.index("<!-- table5-begin -->") + len("<!-- table5-begin -->"), s.index("<!-- table5-end -->")
        open(p, "w").write(s[:a0] + "\n" + text + "\n" + s[b0:])
    else:
        print(text)


if __name__ == "__main__":
    main()
