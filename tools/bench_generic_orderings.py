"""Diagnostic: the generic (arbitrary dofmap) stiffness kernel under hostile
orderings of the same cfg2 mesh: lexicographic, random cell order, random dof
numbering, both.  Not part of the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wave_fenics_amd as w  # noqa: E402


def time_op(op, x, y, reps=15):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.15:     # past the power ramp after idle (profiles/r03_power_ramp.md)
        for _ in range(20):
            op(x, y)
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        op(x, y)
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def main():
    p = int(os.environ.get("P", "4"))
    n = int(os.environ.get("N", "54"))
    dev = torch.device("cuda", 0)
    mesh = w.create_box(n)
    V = w.create_functionspace(mesh, p)
    rng = np.random.default_rng(0)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    cperm = rng.permutation(mesh.ncells)
    dperm = rng.permutation(V.ndofs).astype(np.int32)
    # "first touch": dofs numbered in the order a cell-by-cell traversal meets them (what a dofmap
    # builder that walks the cells produces): local, but not lexicographic
    flat = V.dofmap.reshape(-1)
    _, first = np.unique(flat, return_index=True)
    ft = np.empty(V.ndofs, dtype=np.int32)
    ft[flat[np.sort(first)]] = np.arange(V.ndofs, dtype=np.int32)
    for name, cp, dp in (("lexicographic", None, None), ("random cell order", cperm, None), ("first-touch numbering", None, ft),
                         ("first-touch + random cells", cperm, ft), ("random dof numbering", None, dperm),
                         ("both random", cperm, dperm), ("both random + wf_lattice_numbering", cperm, "lattice")):
        dm = V.dofmap if cp is None else V.dofmap[cp]
        gd = mesh.geom_dofmap if cp is None else mesh.geom_dofmap[cp]
        if isinstance(dp, str):
            # the setup-time renumbering option applied to the scrambled space
            import time as _t
            dm = dperm[dm]
            Vs = w.FunctionSpace(mesh, p, np.ascontiguousarray(dm), w.IndexMap(V.ndofs), V.lattice, structured=False)
            t0 = _t.time()
            new = w.lattice_numbering(Vs)
            name += f" ({_t.time() - t0:.1f} s on the host)"
            dm = new[dm]
        elif dp is not None:
            dm = dp[dm]
        m2 = w.BoxMesh(mesh.n, mesh.x, np.ascontiguousarray(gd))
        V2 = w.FunctionSpace(m2, p, np.ascontiguousarray(dm), w.IndexMap(V.ndofs), V.lattice, structured=False)
        kern = os.environ.get("KERNEL")
        op = w.StiffnessOperator(V2, p, structured=False, tuning={"kernel": kern} if kern else None)
        name = f"{name} [{op.kernel}]"
        t = time_op(op, x, y)
        print(f"{name:44s} {t:8.3f} ms   {op.alg_bytes()/t/1e6:8.0f} GB/s alg   frac {op.alg_bytes()/t/1e6/8000:.3f}", flush=True)
        del op


if __name__ == "__main__":
    main()
