#!/usr/bin/env python3
"""Diagnostic: phase timeline of the indexed marching kernel of the low degrees (k_march_idx, P <= 3; P >= 4 runs
the k-split kernel, the dense mass has tools/mass_trace.py), from a library built with -DWF_IDX_TRACE
(examples/bin/libwavehip_itrace.so; see tools/march_trace.sh for the recipe).
usage: idx_trace.py [P]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wave_fenics_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.environ.get("WAVEHIP_LIB") or os.path.join(ROOT, "examples", "bin", "libwavehip_itrace.so")
import wave_fenics_amd as w   # noqa: E402

ITERS, SLOTS = 12, 6


def main():
    kind = "stiffness"
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = 216 // p
    dev = torch.device("cuda", 0)
    V = w.create_functionspace(w.create_box(n), p, build_dofmap=True)
    V.structured = False
    op = w.StiffnessOperator(V, p, {"c0": 1500.0})
    assert op.kernel == "march_idx" and p <= 3
    names = ["(a) issue prefetch", "(b) element kernels", "tile add + barrier", "rotate", "flush (atomics)", "end barrier"]
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros(V.ndofs, dtype=torch.float64, device=dev)
    for _ in range(3):
        op(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    op(x, y)
    e1.record()
    torch.cuda.synchronize()
    print(f"{kind} P{p}, {V.ndofs} dofs: apply {e0.elapsed_time(e1):.4f} ms (with the timestamp stores)")
    L = _lib.lib()
    buf = np.zeros(512 * 4 * ITERS * SLOTS, dtype=np.uint64)
    L.wf_debug_idx_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.wf_debug_idx_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(512, 4, ITERS, SLOTS).astype(np.float64) * 0.01   # us
    t = np.where(t > 0, t - t[t > 0].min(), np.nan)
    for k in range(5):
        d = t[:, :, 1:, k + 1] - t[:, :, 1:, k]
        print(f"{names[k]:40s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 5]
    print(f"{names[5]:40s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 0]
    print(f"{'whole layer':40s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")


if __name__ == "__main__":
    main()
