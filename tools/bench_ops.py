"""Secondary measurements (not the driver's bench line): every operator of the
path at ~10 M dofs on one MI355X -- BASELINE.json configs[1] (P4 stiffness) and
configs[2] (P6 mass, lumped and dense/TSMM form) plus neighbours.  Prints one
JSON object per line; HIP-event medians, inputs resident in HBM."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wave_fenics_amd as w  # noqa: E402
from wave_fenics_amd import la  # noqa: E402


def timeit(fn, reps=20, warm=3, settle_s=0.4):
    """HIP-event median of `reps` calls at the GPU's sustained operating point: after `warm` calls the
    function is run untimed for `settle_s` seconds, because the first ~60 launches after idle run up to 25 %
    slower while board power ramps (profiles/r03_power_ramp.md).  SETTLE=0 in the environment times from idle."""
    import time
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    settle_s = float(os.environ.get("SETTLE", settle_s))
    t0 = time.time()
    while time.time() - t0 < settle_s:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def report(name, ms, alg_bytes, ndofs, extra=None):
    out = {"op": name, "ms": round(ms, 4), "alg_GBs": round(alg_bytes / ms / 1e6, 1),
           "frac_of_8TBs": round(alg_bytes / ms / 1e6 / 8000, 3), "Gdofs_per_s": round(ndofs / ms / 1e6, 2)}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def gl_rule(m):
    x, wt = np.polynomial.legendre.leggauss(m)
    return 0.5 * (x + 1), 0.5 * wt


def lagrange(nodes, pts):
    n = len(nodes)
    phi = np.ones((len(pts), n))
    for a in range(n):
        for b in range(n):
            if b != a:
                phi[:, a] *= (pts - nodes[b]) / (nodes[a] - nodes[b])
    return phi


def bench_tet(dev):
    import time
    from wave_fenics_amd import tet
    p, n = 4, int(os.environ.get("TET_N", "54"))
    t0 = time.time()
    V = tet.create_kuhn_box(n, p)
    t1 = time.time()
    op = tet.TetStiffnessOperator(V, p)
    t2 = time.time()
    N = V.ndofs
    x = torch.rand(N, dtype=torch.float64, device=dev)
    y = torch.zeros(N, dtype=torch.float64, device=dev)
    ms = timeit(lambda: op(x, y))
    report(f"tet P{p} dense stiffness (MFMA f64 16x16x4), Kuhn box {n}^3 cubes", ms, op.alg_bytes(), N,
           {"cells": V.ncells, "ndofs": N, "TFLOPs_dense_model": round(op.flops() / ms / 1e9, 2),
            "frac_of_f64_mfma_peak_78.6TF": round(op.flops() / ms / 1e9 / 78.6, 3),
            "mesh_s": round(t1 - t0, 2), "setup_s": round(t2 - t1, 2)})


def bench_tsmm(dev):
    # demo/gpu_tsmm/main.cpp: ndofs = 125, ncells = 100000, two products, GFLOPs = 4*ncells*nd^2/t
    for ncells, nd in ((100000, 125), (1000000, 125), (1000000, 64), (2000000, 27), (300000, 216), (200000, 343)):
        xe = torch.rand(ncells * nd, dtype=torch.float64, device=dev)
        xq = torch.zeros_like(xe)
        ue = torch.zeros_like(xe)
        phi = torch.rand(nd, nd, dtype=torch.float64, device=dev)
        for layout in (0, 1):
            def two():
                w.tsmm(ncells, xe, phi, xq, layout=layout)
                w.tsmm(ncells, xq, phi, ue, layout=layout)
            ms = timeit(two)
            print(json.dumps({"op": f"TSMM x2 ({ncells} x {nd}) . ({nd} x {nd}), layout {layout}", "ms": round(ms, 4),
                              "GFLOPs": round(4.0 * ncells * nd * nd / ms / 1e6, 1),
                              "frac_of_f64_mfma_78.6TF": round(4.0 * ncells * nd * nd / ms / 1e9 / 78.6, 3),
                              "GBs": round(4 * 8.0 * ncells * nd / ms / 1e6, 1)}), flush=True)


def main():
    dev = torch.device("cuda", 0)
    only = sys.argv[1:] or ["stiffness", "mass", "dense", "vector"]
    if "tsmm" in only:
        bench_tsmm(dev)
        only = [o for o in only if o != "tsmm"]
        if not only:
            return
    if "tet" in only:
        bench_tet(dev)
        only = [o for o in only if o != "tet"]
        if not only:
            return
    degrees = [int(v) for v in os.environ.get("DEGREES", "2,4,6").split(",")]
    for p in degrees:
        n = {1: 216, 2: 108, 3: 72, 4: 54, 5: 43, 6: 36, 7: 31}[p]     # ~10.2 M dofs each
        mesh = w.create_box(n)
        V = w.create_functionspace(mesh, p, build_dofmap=True)
        N = V.ndofs
        x = torch.rand(N, dtype=torch.float64, device=dev)
        y = torch.zeros(N, dtype=torch.float64, device=dev)
        tag = {"degree": p, "cells": mesh.ncells, "ndofs": N}
        if "stiffness" in only:
            cases = [(True, None), (False, None), (False, {"kernel": "batch"})]
            if "KS" in os.environ and p <= 4:
                cases = [(True, None), (True, {"variant": 3})]
            for structured, tuning in cases:
                op = w.StiffnessOperator(V, p, structured=structured, tuning=tuning)
                report(f"stiffness P{p} " + ("box" if structured else "any dofmap") + f" [{op.kernel}]", timeit(lambda: op(x, y)),
                       op.alg_bytes(), N, dict(tag, kernel=op.kernel, lz=op.info.plan_lz, tuning=str(tuning)))
                del op
        if "mass" in only:
            op = w.SpectralMassOperator(V, p, structured=False)
            report(f"lumped mass P{p} generic (fused gather*detJ->scatter)", timeit(lambda: op(x, y)), op.alg_bytes(), N, tag)
            del op
            op = w.SpectralMassOperator(V, p, structured=True)
            report(f"lumped mass P{p} box (pre-assembled diagonal)", timeit(lambda: op(x, y)), op.alg_bytes(), N, tag)
            del op
        if "dense" in only:
            pts, wts, D = w.tabulate_gll(p)
            # collocated GLL rule (demo/gpu_operator_monolithic) and Gauss rule of degree 2p (demo/gpu_operator)
            for label, (qp, qw), nodes in (("gll-collocated", (pts, wts), pts),
                                           ("equispaced+gauss", gl_rule(p + 1), np.linspace(0, 1, p + 1))):
                phi1 = lagrange(nodes, qp)
                m = len(qp)
                _, detq = w.precompute_geometric_data(mesh, p, use_fabs=False, clamp=False, want_G=False)
                if m != p + 1 or label != "gll-collocated":
                    # affine box: detJ = vol * w_q
                    W3 = np.einsum("k,j,i->kji", qw, qw, qw).reshape(-1)
                    detq = np.tile(W3 / mesh.ncells, (mesh.ncells, 1))
                op = w.MassOperator(V, p, phi1, detq)
                alg = mesh.ncells * (8.0 * m ** 3 + 4.0 * (p + 1) ** 3) + 16.0 * N
                report(f"dense mass P{p} {label} (sum-factorised Phi^T D Phi)", timeit(lambda: op.apply(x, y), reps=10),
                       alg, N, dict(tag, flops_ref_model=op.flops()))
                del op
        if "vector" in only and p == 4:
            z = torch.zeros_like(x)
            report("axpy r=a*x+y", timeit(lambda: la.axpy(z, 0.5, x, y)), 24.0 * N, N)
            report("pointwise_div", timeit(lambda: la.pointwise_div(x, y, z)), 24.0 * N, N)
            report("copy", timeit(lambda: la.copy(x, z)), 16.0 * N, N)
            report("fill", timeit(lambda: la.fill(z, 0.0)), 8.0 * N, N)
        if "rk4" in only and p == 4:
            from wave_fenics_amd.linear_gll import LinearGLLOpt, cfl_time_step
            V.structured = True
            hi = (0.1, 0.1, 0.1)
            mesh2 = w.create_box(n, hi=hi)
            V2 = w.create_functionspace(mesh2, p, build_dofmap=False)
            dt, _ = cfl_time_step(mesh2, p, 1500.0, 0.5e6, CFL=0.25)
            for fused in (False, True):
                eqn = LinearGLLOpt(V2, p, 1500.0, 0.5e6, 6e4)
                eqn.init()
                run = eqn.rk4_fused if fused else eqn.rk4
                nwarm = 100 if float(os.environ.get("SETTLE", "1")) > 0 else 3   # past the power ramp after idle
                run(0.0, nwarm * dt - 1e-13, dt)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                nsteps = 50
                e0.record()
                run(nwarm * dt, (nwarm + nsteps) * dt - 1e-13, dt)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / nsteps
                print(json.dumps({"op": "RK4 time step P4 (4 stages: K + boundary + vector algebra) " + ("fused" if fused else "reference-order"),
                                  "ms_per_step": round(ms, 4), "Gdof_stages_per_s": round(4 * N / ms / 1e6, 2),
                                  "finite": bool(torch.isfinite(eqn.u_n).all())}), flush=True)
                del eqn
        del V, mesh, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
