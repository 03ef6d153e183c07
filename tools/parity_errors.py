"""Print the achieved max relative error of every stiffness kernel against the CPU
oracle (test infrastructure; run on the GPU box).  Used to state tolerances."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wave_fenics_amd as w
from oracle import wave_oracle as o

dev = torch.device("cuda", 0)
for p, n in [(1, (6, 5, 4)), (2, (5, 4, 4)), (3, (5, 3, 3)), (4, (6, 5, 3)), (4, (11, 4, 7)), (5, (3, 2, 2)), (6, (3, 2, 2)), (7, (2, 2, 1))]:
    for perturb in (0.0, 0.2):
        om = o.create_box(n, p, perturb=perturb)
        mesh = w.create_box(n, perturb=perturb)
        V = w.create_functionspace(mesh, p)
        K = o.StiffnessOperator(om, p)
        x = np.random.default_rng(1).uniform(-1, 1, om.ndofs)
        yref = np.zeros(om.ndofs); K(x, yref)
        row = []
        for name, kw in [("march_idx(G)", dict(structured=False, G=K.G, tuning={"kernel": "march"})),
                         ("march_idx", dict(structured=False, tuning={"kernel": "march"})),
                         ("batch", dict(structured=False, tuning={"kernel": "batch"})),
                         ("march_box", dict(structured=True)), ("box_block", dict(structured=True, tuning={"kernel": "box_block"}))]:
            y = torch.zeros(om.ndofs, dtype=torch.float64, device=dev)
            op = w.StiffnessOperator(V, p, {"c0": 1500.0}, **kw)
            op(torch.from_numpy(x).to(dev), y)
            row.append(f"{name}[{op.kernel}] {np.abs(y.cpu().numpy() - yref).max() / np.abs(yref).max():.2e}")
        G, detJ = w.precompute_geometric_data(mesh, p)
        row.append(f"G {np.abs(G - K.G).max() / np.abs(K.G).max():.2e}")
        print(f"P{p} n={n} perturb={perturb}: " + "  ".join(row), flush=True)
