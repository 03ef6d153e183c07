#!/usr/bin/env python3
"""Diagnostic: phase timeline of the box marching kernel (k_stiffness_march) from the trace build of
the library (tools/march_trace.sh): median duration of each phase of a layer over waves and layers."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wave_fenics_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.environ.get("WAVEHIP_LIB") or os.path.join(ROOT, "examples", "bin", "libwavehip_mtrace.so")
import wave_fenics_amd as w   # noqa: E402

ITERS, SLOTS = 12, 6
NAMES = ["(a) issue prefetch", "(b) element kernels (2 phases, 1 barrier)", "O write + barrier", "(c) rotate (waits for the prefetch)",
         "(d) combine + atomics", "end barrier"]


def main():
    n, p = int(os.environ.get("N", "54")), int(os.environ.get("P", "4"))
    dev = torch.device("cuda", 0)
    V = w.create_functionspace(w.create_box(n), p, build_dofmap=False)
    V.structured = True
    K = w.StiffnessOperator(V, p, {"c0": 1500.0})
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros(V.ndofs, dtype=torch.float64, device=dev)
    for _ in range(3):
        K(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K(x, y)
    e1.record()
    torch.cuda.synchronize()
    print(f"apply: {e0.elapsed_time(e1):.4f} ms (with the timestamp stores)")
    L = _lib.lib()
    buf = np.zeros(512 * 4 * ITERS * SLOTS, dtype=np.uint64)
    L.wf_debug_march_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.wf_debug_march_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(512, 4, ITERS, SLOTS).astype(np.float64) * 0.01   # us
    t = np.where(t > 0, t - t[t > 0].min(), np.nan)
    for k in range(5):
        d = t[:, :, 1:, k + 1] - t[:, :, 1:, k]
        print(f"{NAMES[k]:45s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 5]
    print(f"{NAMES[5]:45s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 0]
    print(f"{'whole layer':45s} median {np.nanmedian(d):6.2f} us   p10 {np.nanpercentile(d, 10):6.2f}   p90 {np.nanpercentile(d, 90):6.2f}")
    for b in (0, 1):
        print(f"wg {b} wave 0:")
        for it in range(1, 5):
            print("   " + " ".join(f"{v:8.2f}" for v in t[b, 0, it]))


if __name__ == "__main__":
    main()
