"""Diagnostic: the tetrahedral MFMA kernel under the WF_ABLATE masks (needs the diagnostic build of the
library, tools/diag_build.sh)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ.setdefault("WAVEHIP_LIB", os.path.join(ROOT, "examples", "bin", "libwavehip_diag.so"))
import numpy as np, torch
from wave_fenics_amd import tet
dev = torch.device("cuda", 0)
V = tet.create_kuhn_box(54, 4)
op = tet.TetStiffnessOperator(V, 4)
x = torch.rand(V.ndofs, dtype=torch.float64, device=dev); y = torch.zeros_like(x)
def t(reps=10):
    for _ in range(2): op(x, y)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); op(x, y); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for m in (0, 1, 4, 5, 8, 9, 12, 13):
    os.environ["WF_ABLATE"] = str(m)
    print(m, round(t(), 4), flush=True)
