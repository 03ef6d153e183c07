"""Per-launch duration of one stiffness operator over a long back-to-back run, with the GPU's clocks
and power sampled beside it (rocm-smi, read only): shows whether a kernel's time under sustained load
is set by the power-managed shader clock rather than by the launch itself.
  python tools/drift.py [P] [launches] [gap_us]"""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wave_fenics_amd as w  # noqa: E402

NCELL = {2: 108, 3: 72, 4: 54, 5: 43, 6: 36, 7: 31}


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = next(iter(d.values()))
        return {k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "power"))}
    except Exception as e:  # diagnostics only
        return {"error": str(e)}


def main():
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device("cuda", 0)
    V = w.create_functionspace(w.create_box(NCELL[p]), p, build_dofmap=False)
    op = w.StiffnessOperator(V, p, structured=True)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    for _ in range(3):
        op(x, y)
    torch.cuda.synchronize()
    print(json.dumps({"idle": smi()}), flush=True)
    time.sleep(1.0)
    samples = []
    stop = False

    def sampler():
        while not stop:
            samples.append(smi())
            time.sleep(0.05)

    th = threading.Thread(target=sampler)
    th.start()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    rounds = 0
    t0 = time.time()
    while time.time() - t0 < 3.0:     # keep the load on long enough for rocm-smi to see it
        for a, b in ev:
            a.record()
            op(x, y)
            b.record()
            if gap:
                torch.cuda._sleep(int(gap * 2100))
        torch.cuda.synchronize()
        t = np.array([a.elapsed_time(b) for a, b in ev])
        if rounds == 0:
            first = t.copy()
        rounds += 1
    stop = True
    th.join()
    print(json.dumps({"P": p, "kernel": op.kernel, "gap_us": gap, "first_round_us": [int(v * 1000) for v in first[:60]],
                      "first_round_median": round(float(np.median(first)), 4), "last_round_median": round(float(np.median(t)), 4),
                      "last_round_min": round(float(t.min()), 4), "rounds": rounds}), flush=True)
    for s in samples[:3] + samples[-3:]:
        print(json.dumps({"loaded": s}), flush=True)


if __name__ == "__main__":
    main()
