#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory (gpurun_out/prof_<tag>) into
profiles/<tag>_*.  Applies the gfx950 counter corrections of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are in KiB; FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams
(the 16-B/lane geometry loads here), so the read side is doubled; WRITE_SIZE is
exact for 16-B streaming stores and float atomics."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    kernel_key = sys.argv[2] if len(sys.argv) > 2 else "k_stiffness_march"
    alg_bytes = float(sys.argv[3]) if len(sys.argv) > 3 else 1187009008.0
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = [f"# rocprofv3 summary {tag}", "",
             f"source: `tools/profile.sh {tag}` (bench.py, cfg2: P4, 54^3 cells, 10 218 313 dofs, 1 MI355X)", ""]
    stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    avg_ns = None
    if stats:
        lines += ["## rocprofv3 --kernel-trace --stats", "",
                  "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
        with open(stats[0]) as f:
            for r in csv.DictReader(f):
                nm = short(r["Name"])
                if nm.startswith("at::") and float(r["Percentage"]) < 1.0:
                    continue
                lines.append(f"| `{nm[:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | "
                             f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
                if kernel_key in nm:
                    avg_ns = float(r["AverageNs"])
        lines.append("")
        with open(stats[0]) as f, open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    counters = defaultdict(list)
    meta = {}
    for cc in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        with open(cc) as f:
            for r in csv.DictReader(f):
                if kernel_key in r["Kernel_Name"]:
                    counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                              "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    pm = {k: sum(v) / len(v) for k, v in counters.items()}
    if meta:
        lines += ["## dispatch of `" + kernel_key + "`", "", ", ".join(f"{k}={v}" for k, v in meta.items()), ""]
    lines += [f"## PMC per launch of `{kernel_key}` (mean over launches, one pass per counter group)", "",
              "| counter | value |", "|---|---|"]
    for k in sorted(pm):
        lines.append(f"| {k} | {pm[k]:.6g} |")
    lines.append("")
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        fetch_raw = pm["FETCH_SIZE"] * 1024.0
        write = pm["WRITE_SIZE"] * 1024.0
        traffic = 2.0 * fetch_raw + write
        out = {"stiffness_hbm_bytes_per_launch": traffic, "fetch_bytes_corrected": 2.0 * fetch_raw,
               "fetch_bytes_raw": fetch_raw, "write_bytes": write, "tag": tag,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; "
                         "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B)"}
        lines += ["## HBM traffic of the stiffness kernel", "",
                  f"- FETCH_SIZE raw {fetch_raw/1e6:.1f} MB -> corrected x2 = {2*fetch_raw/1e6:.1f} MB",
                  f"- WRITE_SIZE {write/1e6:.1f} MB",
                  f"- traffic per launch = {traffic/1e6:.1f} MB (algorithmic contract figure {alg_bytes/1e6:.1f} MB)", ""]
        if avg_ns:
            lines.append(f"- rocprof average duration {avg_ns/1e3:.1f} us -> {alg_bytes/avg_ns:.0f} GB/s algorithmic "
                         f"({alg_bytes/avg_ns/80:.1f} % of 8 TB/s), {traffic/avg_ns:.0f} GB/s measured traffic")
            lines.append("")
        with open(os.path.join(dst, "traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
    if "SQ_LDS_BANK_CONFLICT" in pm and "SQ_LDS_IDX_ACTIVE" in pm:
        lines.append(f"- LDS bank-conflict cycles / LDS active cycles = "
                     f"{pm['SQ_LDS_BANK_CONFLICT']/max(pm['SQ_LDS_IDX_ACTIVE'],1):.3f}")
    if "TCC_HIT_sum" in pm and "TCC_MISS_sum" in pm:
        lines.append(f"- L2 hit rate = {pm['TCC_HIT_sum']/(pm['TCC_HIT_sum']+pm['TCC_MISS_sum']):.3f} "
                     "(atomics count as misses)")
    if "SQ_WAVE_CYCLES" in pm:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in pm:
                lines.append(f"- {k} / SQ_WAVE_CYCLES = {pm[k]/pm['SQ_WAVE_CYCLES']:.3f}")
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
