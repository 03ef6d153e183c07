#!/usr/bin/env python3
"""Turn a tools/evidence.sh output directory (gpurun_out/ev_<tag>) into the tracked files
profiles/<tag>_*.  Counter corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM
section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE tallies the 128-B requests of wide
coalesced streams (the 16-B/lane geometry loads) at 64 B, so the read side is doubled;
WRITE_SIZE is exact for 16-B streaming stores and float atomics."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def stats_table(path, only_wf=False, top=30):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            nm = short(r["Name"])
            if only_wf and not nm.startswith("wf::") and "rccl" not in nm:
                continue
            if nm.startswith("at::") and float(r["Percentage"]) < 1.0:
                continue
            rows.append((nm, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                         float(r["Percentage"])))
    out = ["| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for nm, c, a, mi, ma, p in rows[:top]:
        out.append(f"| `{nm[:72]}` | {c} | {a:.2f} | {mi:.2f} | {ma:.2f} | {p:.2f} |")
    return out, {nm: a for nm, c, a, mi, ma, p in rows}


def first(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None


def pmc_means(src, prefix, key):
    counters, meta = defaultdict(list), {}
    for cc in glob.glob(os.path.join(src, prefix + "*", "*", "*_counter_collection.csv")):
        with open(cc) as f:
            for r in csv.DictReader(f):
                if key in r["Kernel_Name"]:
                    counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                              "Accum_VGPR_Count", "SGPR_Count") if k in r}
    return {k: sum(v) / len(v) for k, v in counters.items()}, meta


def traffic_block(pm, alg, must, avg_us, label):
    lines, out = [], None
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        fr, wr = pm["FETCH_SIZE"] * 1024.0, pm["WRITE_SIZE"] * 1024.0
        tr = 2.0 * fr + wr
        out = {"hbm_bytes_per_launch": tr, "fetch_bytes_corrected": 2.0 * fr, "fetch_bytes_raw": fr, "write_bytes": wr}
        lines += [f"- FETCH_SIZE raw {fr/1e6:.1f} MB -> corrected x2 = {2*fr/1e6:.1f} MB; WRITE_SIZE {wr/1e6:.1f} MB",
                  f"- traffic per launch = **{tr/1e6:.1f} MB** (algorithmic contract figure {alg/1e6:.1f} MB"
                  + (f", bytes the kernel must move {must/1e6:.1f} MB" if must else "") + ")"]
        if avg_us:
            lines.append(f"- {label} average duration {avg_us:.1f} us -> **{alg/avg_us/1e3:.0f} GB/s algorithmic = "
                         f"{alg/avg_us/1e3/80:.1f} % of 8 TB/s**"
                         + (f" ({must/avg_us/1e3/80:.1f} % on the bytes it must move)" if must else "")
                         + f"; measured traffic / time = {tr/avg_us/1e3:.0f} GB/s")
    return lines, out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", f"ev_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    L = [f"# rocprofv3 / bench evidence {tag}", "",
         f"source: `tools/evidence.sh {tag}` on one MI355X (gfx950, ROCm 7.2); summarised by `tools/summarise_evidence.py`.",
         "cfg2 = BASELINE.json configs[1]: P4, 54^3 cells, 10 218 313 dofs.", ""]
    # ---- bench lines
    for nm in ("bench_line", "bench_generic", "bench_periodic_x", "bench_periodic_xyz", "bench_generic_periodic_xyz"):
        p = os.path.join(src, nm + ".json")
        if os.path.exists(p):
            txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if txt:
                with open(os.path.join(dst, f"{tag}_{nm}.json"), "w") as f:
                    f.write(txt[-1] + "\n")
                d = json.loads(txt[-1])
                r = d["roofline"]
                L.append(f"- `{nm}`: {d['ms_per_step']:.4f} ms/step, {d['value']/1e9:.2f} Gdof/s; stiffness kernel "
                         f"{r['kernel_ms']:.4f} ms = {r['achieved']:.0f} GB/s algorithmic = {r['frac']:.3f} of 8 TB/s"
                         + (f"; cpu_baseline {d['cpu_baseline']['value']/1e6:.2f} Mdof/s on {d['cpu_baseline']['cores']} threads"
                            if "cpu_baseline" in d and d["cpu_baseline"].get("value") else ""))
    L.append("")
    alg = 1187009008.0
    must_box = alg - 4.0 * 125 * 157464
    traffic = {}
    for kind, key, must in (("box", "k_stiffness_march<", must_box), ("generic", "k_march_idx<", alg)):
        st = first(os.path.join(src, f"kt_{kind}", "*", "*_kernel_stats.csv"))
        if not st:
            continue
        shutil.copy(st, os.path.join(dst, f"{tag}_{kind}_kernel_stats.csv"))
        tbl, avgs = stats_table(st)
        L += [f"## bench.py{' --generic' if kind == 'generic' else ''}: rocprofv3 --kernel-trace --stats", ""] + tbl + [""]
        avg = next((a for n, a in avgs.items() if key.replace("wf::", "") in n), None)
        pm, meta = pmc_means(src, f"pmc_{kind}_", key)
        if meta:
            L += [f"dispatch of `{key}...>`: " + ", ".join(f"{k}={v}" for k, v in meta.items()), ""]
        if pm:
            L += ["| counter (mean per launch) | value |", "|---|---|"] + [f"| {k} | {pm[k]:.6g} |" for k in sorted(pm)] + [""]
        tl, tr = traffic_block(pm, alg, must if kind == "box" else 0.0, avg, "rocprof")
        L += tl + [""]
        if tr:
            traffic[kind] = tr
        if "SQ_LDS_BANK_CONFLICT" in pm and "SQ_LDS_IDX_ACTIVE" in pm:
            L.append(f"- LDS bank-conflict cycles / LDS active cycles = {pm['SQ_LDS_BANK_CONFLICT']/max(pm['SQ_LDS_IDX_ACTIVE'],1):.3f}")
        if "TCC_HIT_sum" in pm and "TCC_MISS_sum" in pm:
            L.append(f"- L2 hit rate = {pm['TCC_HIT_sum']/(pm['TCC_HIT_sum']+pm['TCC_MISS_sum']):.3f} (atomics count as misses)")
        if "SQ_WAVE_CYCLES" in pm:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in pm:
                    L.append(f"- {k} / SQ_WAVE_CYCLES = {pm[k]/pm['SQ_WAVE_CYCLES']:.3f}")
        L.append("")
    if "box" in traffic:
        import hashlib
        with open(os.path.join(ROOT, "wave_fenics_amd", "libwavehip.so"), "rb") as f:
            lib_sha = hashlib.sha256(f.read()).hexdigest()[:16]
        out = {"stiffness_hbm_bytes_per_launch": traffic["box"]["hbm_bytes_per_launch"], **traffic["box"], "tag": tag,
               "lib_sha16": lib_sha,   # bench.py quotes this figure only for the library build it was measured on
               "generic_kernel": traffic.get("generic"),
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE x2 "
                         "(gfx950 tallies 128-B requests at 64 B)"}
        with open(os.path.join(dst, "traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
    # ---- all operators
    p = os.path.join(src, "ops.jsonl")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_ops.jsonl"))
        L += ["## every operator at ~10 M dofs (`tools/bench_ops.py`, HIP-event medians)", "", "| op | ms | fraction |", "|---|---|---|"]
        for l in open(p):
            if not l.startswith("{"):
                continue
            d = json.loads(l)
            ms = d.get("ms", d.get("ms_per_step"))
            fr = d.get("frac_of_8TBs")
            fm = d.get("frac_of_f64_mfma_78.6TF", d.get("frac_of_f64_mfma_peak_78.6TF"))
            L.append(f"| {d['op']} | {ms} | " + (f"{fm} of 78.6 TF f64 MFMA" if fm is not None else (f"{fr} of 8 TB/s" if fr is not None else "")) + " |")
        L.append("")
    # ---- FETCH / WRITE / LDS / wait counters of the other kernels (separate passes each)
    groups = (("pmc_ops6_", "P6 stiffness and dense mass (tools/bench_ops.py, DEGREES=6)"), ("pmc_rk4_", "RK4 loop (tools/bench_rk4.py)"),
              ("pmc_tet_", "tetrahedral kernel (tools/bench_ops.py tet)"))
    for prefix, title in groups:
        agg = defaultdict(lambda: defaultdict(list))
        for cc in glob.glob(os.path.join(src, prefix + "*", "*", "*_counter_collection.csv")):
            with open(cc) as f:
                for r in csv.DictReader(f):
                    nm = short(r["Kernel_Name"])
                    if nm.startswith("wf::") and "geometry" not in nm:
                        agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not agg:
            continue
        L += [f"## counters: {title}", "",
              "traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 tallies the 128-B requests of wide streams at 64 B); "
              "conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wait / active = share of SQ_WAVE_CYCLES.", "",
              "| kernel | launches | traffic MB | write MB | LDS conflict | wait | issue stall | active |", "|---|---|---|---|---|---|---|---|"]
        for k, c in sorted(agg.items()):
            m = {kk: sum(v) / len(v) for kk, v in c.items()}
            tr = (2.0 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024.0 / 1e6
            wc = m.get("SQ_WAVE_CYCLES", 0)
            fmt = lambda a, b: f"{a / b:.3f}" if b else ""
            L.append(f"| `{k[:64]}` | {len(next(iter(c.values())))} | {tr:.1f} | {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} | "
                     f"{fmt(m.get('SQ_LDS_BANK_CONFLICT', 0), m.get('SQ_LDS_IDX_ACTIVE', 0))} | {fmt(m.get('SQ_WAIT_ANY', 0), wc)} | "
                     f"{fmt(m.get('SQ_WAIT_INST_ANY', 0), wc)} | {fmt(m.get('SQ_ACTIVE_INST_ANY', 0), wc)} |")
        L.append("")
    p = os.path.join(src, "shapes.jsonl")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_shapes.jsonl"))
        L += ["## k-split kernel, every compiled cross-section (`tools/bench_shapes.py`)", "", "| P | configuration | ms | of 8 TB/s |", "|---|---|---|---|"]
        for l in open(p):
            if l.startswith("{"):
                d = json.loads(l)
                L.append(f"| {d['P']} | {d['cfg']} (lz {d['lz']}) | {d['ms']} | {d['frac_8TBs']} |")
        L.append("")
    for nm, title in (("kt_ops", "operators"), ("kt_mfma", "TSMM and tetrahedral kernels"), ("kt_rk4", "RK4 loop (tools/bench_rk4.py, fused)"),
                      ("kt_rk4_periodic", "RK4 loop, periodic xyz partition on one rank (RCCL exchange with itself)")):
        st = first(os.path.join(src, nm, "*", "*_kernel_stats.csv"))
        if st:
            shutil.copy(st, os.path.join(dst, f"{tag}_{nm[3:]}_kernel_stats.csv"))
            tbl, _ = stats_table(st, only_wf=True)
            L += [f"## kernel stats: {title}", ""] + tbl + [""]
    # ---- MFMA pipe utilisation
    cc = first(os.path.join(src, "pmc_mfma", "*", "*_counter_collection.csv"))
    if cc:
        agg = defaultdict(lambda: defaultdict(list))
        with open(cc) as f:
            for r in csv.DictReader(f):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        L += ["## fp64 MFMA pipe utilisation (`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, own pass)", "",
              "utilisation = MFMA busy cycles (summed over 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024).", "",
              "| kernel | launches | MFMA busy | GRBM_GUI_ACTIVE | utilisation |", "|---|---|---|---|---|"]
        for k, c in agg.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0:
                b = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
                g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
                L.append(f"| `{k[:60]}` | {len(c['GRBM_GUI_ACTIVE'])} | {b:.4g} | {g:.4g} | {b/(g/8*1024):.2f} |")
        L.append("")
    for nm in ("rk4.jsonl", "generic_orderings.log", "mfma_f64_rate.log", "hbm_read_rate.log", "tsmm_trace.log", "dense_trace.log", "mass_trace.log", "power_ramp.log"):
        p = os.path.join(src, nm)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_{nm}"))
            L += [f"## {nm}", "", "```"] + [l.rstrip() for l in open(p) if l.strip() and "amdgpu.ids" not in l] + ["```", ""]
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write("\n".join(L) + "\n")
    print("\n".join(L[:60]))


if __name__ == "__main__":
    main()
