"""Tuning sweep of the dense-mass marching kernel (mass_march.hip): layers per work item."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wave_fenics_amd as w

def timeit(fn, reps=12, warm=3):
    import time
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 0.1:      # past the power ramp after idle (profiles/r03_power_ramp.md)
        for _ in range(20): fn()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

dev = torch.device("cuda", 0)
for p in [int(v) for v in os.environ.get("DEGREES", "4,6").split(",")]:
    n = {2: 108, 3: 72, 4: 54, 5: 43, 6: 36, 7: 31}[p]
    mesh = w.create_box(n)
    V = w.create_functionspace(mesh, p)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev); y = torch.zeros_like(x)
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ.get("SHAPES", "0,0").split(";")]
    for (bx, by) in shapes:
        for lz in [int(v) for v in os.environ.get("LZ", "0,1,2,4,8,16").split(",")]:
            tun = {"lz": lz}
            if bx > 0:
                tun["block"] = (bx, by, 1)
            op = w.MassOperator(V, p, variant="equispaced", quad="gauss_jacobi", qdegree=2 * p, tuning=tun)
            ms = timeit(lambda: op.apply(x, y))
            print(json.dumps({"P": p, "block": [bx, by], "lz_req": lz, "lz": op.info.plan_lz, "items": op.info.plan_items,
                              "kernel": op.kernel, "ms": round(ms, 4)}), flush=True)
            del op
