#!/usr/bin/env python3
"""Diagnostic: phase timeline of the dense-mass marching kernel (k_mass_march) from the trace build of the
library (tools/mass_trace.sh): median duration of each phase of a layer over waves and layers.
  P=6 python tools/mass_trace.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wave_fenics_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.environ.get("WAVEHIP_LIB") or os.path.join(ROOT, "examples", "bin", "libwavehip_mstrace.so")
import wave_fenics_amd as w   # noqa: E402

ITERS, SLOTS = 12, 10
NAMES = ["(a) issue x / detJ prefetch", "flush slot 0 + pass X", "flush slot 1 + pass Y", "flush slot 2 + pass Z (fwd, detJ, transposed)",
         "flush slot 3 + pass Y^T", "flush slot 4 + pass X^T (LDS adds) + carry write", "(c) next x planes -> LDS (waits for the prefetch)", "barrier"]
NCELL = {2: 108, 3: 72, 4: 54, 5: 43, 6: 36, 7: 31}


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


def main():
    p = int(os.environ.get("P", "6"))
    n = NCELL[p]
    dev = torch.device("cuda", 0)
    mesh = w.create_box(n)
    V = w.create_functionspace(mesh, p, build_dofmap=True)
    m = p + 1
    q, wq = gl_rule(m)
    phi1 = lagrange(np.linspace(0, 1, p + 1), q)
    W3 = (wq[:, None, None] * wq[None, :, None] * wq[None, None, :]).reshape(-1)
    detq = np.tile(W3 / mesh.ncells, (mesh.ncells, 1))
    op = w.MassOperator(V, p, phi1, detq)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros(V.ndofs, dtype=torch.float64, device=dev)
    for _ in range(3):
        op.apply(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    op.apply(x, y)
    e1.record()
    torch.cuda.synchronize()
    print(f"P{p} kernel {op.kernel} lz {op.info.plan_lz} items {op.info.plan_items}: apply {e0.elapsed_time(e1):.4f} ms (with the timestamp stores)")
    L = _lib.lib()
    buf = np.zeros(512 * 4 * ITERS * SLOTS, dtype=np.uint64)
    L.wf_debug_mass_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.wf_debug_mass_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(512, 4, ITERS, SLOTS).astype(np.float64) * 0.01   # us
    t = np.where(t > 0, t - t[t > 0].min(), np.nan)
    for k in range(len(NAMES)):
        d = t[:, :, 1:, k + 1] - t[:, :, 1:, k]
        print(f"{NAMES[k]:52s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 0]
    print(f"{'whole layer':52s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    for b in (0, 1):
        print(f"wg {b} wave 0:")
        for it in range(1, 4):
            print("   " + " ".join(f"{v:8.2f}" for v in t[b, 0, it]))


if __name__ == "__main__":
    main()
