"""Diagnostic: time the cfg2 stiffness apply under the WF_ABLATE masks and box
block shapes (interleaved rounds in one process).  Not part of the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the WF_ABLATE masks exist only in the diagnostic build of the library (tools/diag_build.sh)
os.environ.setdefault("WAVEHIP_LIB", os.path.join(ROOT, "examples", "bin", "libwavehip_diag.so"))
import wave_fenics_amd as w  # noqa: E402


def time_op(op, x, y, reps=20):
    for _ in range(3):
        op(x, y)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        op(x, y)
        b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    return float(np.median(ts)), float(np.min(ts))


def main():
    p = int(os.environ.get("P", "4"))
    n = int(os.environ.get("N", "54"))
    dev = torch.device("cuda", 0)
    mesh = w.create_box(n)
    V = w.create_functionspace(mesh, p)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros(V.ndofs, dtype=torch.float64, device=dev)
    ops = {}
    ops["generic"] = w.StiffnessOperator(V, p, structured=False)
    ops["batch"] = w.StiffnessOperator(V, p, structured=False, tuning={"kernel": "batch"})
    for spec in sys.argv[1:]:
        # block:bx,by,bz   single-pass block kernel;  march:variant[:lz]   one-thread-per-column marching kernel
        # (variant 0..2) or the k-split kernel (variant 3);  ks:bx,by[:lz]   k-split kernel with that cross-section
        kind, _, rest = spec.partition(":")
        if kind == "block":
            tuning = {"kernel": "box_block", "block": tuple(int(v) for v in rest.split(","))}
        elif kind == "ks":
            shape, _, lz = rest.partition(":")
            bx, by = (int(v) for v in shape.split(","))
            tuning = {"variant": 3, "block": (bx, by, 1), "lz": int(lz or 0)}
        else:
            var, _, lz = rest.partition(":")
            tuning = {"variant": int(var or 0), "lz": int(lz or 0)}
        ops[spec] = w.StiffnessOperator(V, p, structured=True, tuning=tuning)
    alg = ops["generic"].alg_bytes()
    masks = [int(v) for v in os.environ.get("MASKS", "0,1,2,4,8,3,5,7,15").split(",")]
    print(f"P{p} N{n} ndofs {V.ndofs} alg_bytes {alg/1e6:.1f} MB")
    print("kernel".ljust(20) + "".join(f"{m:>9d}" for m in masks))
    for name, op in ops.items():
        row = []
        for m in masks:
            os.environ["WF_ABLATE"] = str(m)
            med, mn = time_op(op, x, y)
            row.append(med)
        os.environ["WF_ABLATE"] = "0"
        print(name.ljust(20) + "".join(f"{t:9.3f}" for t in row), flush=True)
    # plain HBM copy reference on the same device (same byte count as alg_bytes)
    nb = int(alg // 16)
    a = torch.empty(nb, dtype=torch.float64, device=dev)
    b = torch.empty(nb, dtype=torch.float64, device=dev)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"torch copy of {alg/2/1e6:.0f} MB (read+write = alg bytes): {t:.3f} ms = {alg/t/1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
