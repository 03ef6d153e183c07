import sys, time, torch
sys.path.insert(0, "/root/repo")
import wave_fenics_amd as w
from wave_fenics_amd.linear_gll import LinearGLLOpt, cfl_time_step
dev = torch.device("cuda", 0)
for p, n in ((2, 18), (4, 24), (4, 54)):
    mesh = w.create_box(n, hi=(0.1, 0.1, 0.1))
    V = w.create_functionspace(mesh, p, build_dofmap=False)
    dt, _ = cfl_time_step(mesh, p, 1500.0, 0.5e6, CFL=0.25)
    res = {}
    for rnd in range(3):
        for fold in (True, False):
            eqn = LinearGLLOpt(V, p, 1500.0, 0.5e6, 6e4, device=dev)
            eqn.fold_boundary = fold
            eqn.init()
            eqn.rk4_fused(0.0, 5 * dt - 1e-13, dt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ns = 40
            eqn.rk4_fused(5 * dt, (5 + ns) * dt - 1e-13, dt)
            torch.cuda.synchronize()
            res.setdefault(fold, []).append((time.perf_counter() - t0) / ns * 1e3)
    print(p, n, V.ndofs, "fold", min(res[True]), "separate", min(res[False]), flush=True)
