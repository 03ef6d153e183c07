"""Tuning sweep of the k-split marching stiffness kernel (stiffness_march_ks.hip): every compiled column
cross-section of each degree at ~10 M dofs, box and arbitrary-dofmap addressing, optional layers per
z segment.  One JSON line per configuration (HIP-event median of 20 applies, interleaved rounds)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wave_fenics_amd as w  # noqa: E402

SHAPES = {1: [(8, 8), (8, 4)], 2: [(7, 4), (7, 2), (5, 5)], 3: [(4, 4), (4, 2), (2, 2)], 4: [(5, 1), (5, 2), (2, 2), (3, 1)],
          5: [(7, 1), (3, 1), (2, 1)], 6: [(5, 1), (2, 1), (1, 1)], 7: [(2, 2), (2, 1), (1, 1)]}
NCELL = {1: 216, 2: 108, 3: 72, 4: 54, 5: 43, 6: 36, 7: 31}


def main():
    dev = torch.device("cuda", 0)
    degrees = [int(v) for v in os.environ.get("DEGREES", "4,5,6,7").split(",")]
    lzs = [int(v) for v in os.environ.get("LZ", "0").split(",")]
    modes = os.environ.get("MODES", "box").split(",")
    old = "OLD" in os.environ
    for p in degrees:
        n = NCELL[p]
        mesh = w.create_box(n)
        V = w.create_functionspace(mesh, p, build_dofmap="idx" in modes)
        N = V.ndofs
        x = torch.rand(N, dtype=torch.float64, device=dev)
        y = torch.zeros(N, dtype=torch.float64, device=dev)
        ops = []
        for mode in modes:
            for (bx, by) in SHAPES[p]:
                for lz in lzs:
                    t = {"block": (bx, by, 1), "lz": lz}
                    if p == 4 and mode == "box":
                        t["variant"] = 3      # the P4 box operator defaults to k_stiffness_march; 3 selects the k-split kernel
                    ops.append((f"{mode} {bx}x{by} lz={lz}", w.StiffnessOperator(V, p, structured=mode == "box", tuning=t)))
            if old and mode == "box":
                for v in (0, 1, 2):
                    ops.append((f"box old variant {v}", w.StiffnessOperator(V, p, structured=True, tuning={"variant": v})))
        times = {name: [] for name, _ in ops}
        for name, op in ops:
            for _ in range(3):
                op(x, y)
        t0 = time.time()
        while time.time() - t0 < 0.15:     # hold the load past the power ramp after idle (profiles/r03_power_ramp.md)
            for _ in range(20):
                ops[0][1](x, y)
            torch.cuda.synchronize()
        for r in range(5):
            for name, op in ops:
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
                for a, b in ev:
                    a.record()
                    op(x, y)
                    b.record()
                torch.cuda.synchronize()
                times[name] += [a.elapsed_time(b) for a, b in ev]
        for name, op in ops:
            ms = float(np.median(times[name]))
            print(json.dumps({"P": p, "cfg": name, "kernel": op.kernel, "lz": op.info.plan_lz, "items": op.info.plan_items,
                              "ms": round(ms, 4), "min_ms": round(float(np.min(times[name])), 4),
                              "frac_8TBs": round(op.alg_bytes() / ms / 1e6 / 8000, 3)}), flush=True)
        del ops, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
