import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import wave_fenics_amd as w
from wave_fenics_amd._lib import WF_PART_INTERFACE, WF_PART_INTERIOR_A, WF_PART_INTERIOR_B, WF_PART_INTERIOR
dev = torch.device("cuda", 0)
mesh = w.create_box(54); V = w.create_functionspace(mesh, 4, build_dofmap=False)
op = w.StiffnessOperator(V, 4)
x = torch.rand(V.ndofs, dtype=torch.float64, device=dev); y = torch.zeros_like(x)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
print("full", timeit(lambda: op(x, y)))
for g in ((1,1,1),(1,0,0),(0,0,0)):
    op.set_ghost_faces(*[bool(v) for v in g])
    def three():
        op.apply_part(x, y, WF_PART_INTERIOR_A); op.apply_part(x, y, WF_PART_INTERFACE); op.apply_part(x, y, WF_PART_INTERIOR_B)
    def two():
        op.apply_part(x, y, WF_PART_INTERIOR); op.apply_part(x, y, WF_PART_INTERFACE)
    print(g, "items", op.info.items_interior, op.info.items_interface, "3-part", timeit(three), "2-part", timeit(two))
from wave_fenics_amd.distributed import create_distributed_box, VectorUpdater, overlapped_apply
part = create_distributed_box(54, 4, 1, 0)
vu = VectorUpdater(part, device=dev)
for g in ((1,1,1),(1,0,0)):
    op.set_ghost_faces(*[bool(v) for v in g])
    print(g, "two-stream overlapped (no comm)", timeit(lambda: overlapped_apply(op, vu, x, y)))
