#!/usr/bin/env python3
"""Diagnostic: phase timeline of the tetrahedral MFMA kernel (k_stiffness_dense) from the trace
build of the library (tools/dense_trace.sh).  Prints, per phase, the median duration over waves and
batches, and the raw timeline of a few workgroups."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wave_fenics_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.environ.get("WAVEHIP_LIB") or os.path.join(ROOT, "examples", "bin", "libwavehip_trace.so")
from wave_fenics_amd import tet   # noqa: E402

ITERS, SLOTS = 12, 6
NAMES = ["gather ub (LDS)", "issue next loads + MFMA section", "LDS scatter-add", "wait at barrier", "global atomics + refill",
         "top barrier (next batch)"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    dev = torch.device("cuda", 0)
    V = tet.create_kuhn_box(n, 4)
    op = tet.TetStiffnessOperator(V, 4)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros(V.ndofs, dtype=torch.float64, device=dev)
    for _ in range(3):
        op(x, y)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    op(x, y)
    ev1.record()
    torch.cuda.synchronize()
    print(f"apply: {ev0.elapsed_time(ev1):.4f} ms")
    L = _lib.lib()
    buf = np.zeros(512 * 4 * ITERS * SLOTS + 512 * 4, dtype=np.uint64)
    L.wf_debug_dense_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = L.wf_debug_dense_trace(buf.ctypes.data, buf.size)
    assert rc == 0
    hw = buf[512 * 4 * ITERS * SLOTS:].reshape(512, 4)
    buf = buf[:512 * 4 * ITERS * SLOTS]
    hwid = (hw & 0xffffffff).astype(np.int64)
    xcc = ((hw >> 32) & 0xf).astype(np.int64)
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    simd = (hwid >> 4) & 0x3
    place = xcc * 1000 + se * 100 + sh * 10 + cu   # physical CU key
    t = buf.reshape(512, 4, ITERS, SLOTS).astype(np.float64) * 0.01   # us
    t0 = t[t > 0].min()
    t = np.where(t > 0, t - t0, np.nan)
    # phase durations: slot k+1 - slot k within an iteration; top barrier = next iteration's slot 0 - slot 5
    for k in range(5):
        d = t[:, :, 1:, k + 1] - t[:, :, 1:, k]
        print(f"{NAMES[k]:32s} median {np.nanmedian(d):7.2f} us   p10 {np.nanpercentile(d, 10):7.2f}   p90 {np.nanpercentile(d, 90):7.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 5]
    print(f"{NAMES[5]:32s} median {np.nanmedian(d):7.2f} us   p10 {np.nanpercentile(d, 10):7.2f}   p90 {np.nanpercentile(d, 90):7.2f}")
    d = t[:, :, 2:, 0] - t[:, :, 1:-1, 0]
    print(f"{'whole batch':32s} median {np.nanmedian(d):7.2f} us   p10 {np.nanpercentile(d, 10):7.2f}   p90 {np.nanpercentile(d, 90):7.2f}")
    keys = {}
    for b in range(512):
        keys.setdefault(int(place[b, 0]), []).append(b)
    sizes = sorted(len(v) for v in keys.values())
    print(f"distinct CUs: {len(keys)}, workgroups per CU: min {sizes[0]} max {sizes[-1]}")
    shown = 0
    for k, v in keys.items():
        if len(v) >= 2 and shown < 3:
            shown += 1
            print(f"CU key {k}: workgroups {v}, simd of wave 0: {[int(simd[b, 0]) for b in v]}")
            for b in v[:2]:
                for it in range(1, 4):
                    print(f"   wg {b:3d}: " + " ".join(f"{x:8.2f}" for x in t[b, 0, it]))
    for b in ():
        for w in (0,):
            print(f"wg {b} wave {w}:")
            for it in range(1, 5):
                print("   " + " ".join(f"{v:8.2f}" for v in t[b, w, it]))


if __name__ == "__main__":
    main()
