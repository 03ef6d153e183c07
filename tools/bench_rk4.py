#!/usr/bin/env python3
"""Secondary benchmark (not the driver's line): the full RK4 loop of
BASELINE.json configs[3] -- P4 box, 54^3 cells per GPU, domain-decomposed, RCCL
ghost exchange, LinearGLLOpt.rk4_fused -- launched like bench.py:

  python tools/bench_rk4.py [--steps K] [--size N]                      (1 GPU)
  python -m torch.distributed.run --nproc-per-node N tools/bench_rk4.py (N GPUs)

Prints one JSON line: ms per RK4 time step and dof-stages/s over all ranks."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=150, help="untimed steps; long enough to pass the power ramp after idle (profiles/r03_power_ramp.md)")
    ap.add_argument("--size", type=int, default=54)
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--unfused", action="store_true")
    ap.add_argument("--periodic", default="", help="axes identified with their opposite face (one rank: RCCL exchange with itself)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import wave_fenics_amd as w
    from wave_fenics_amd.distributed import VectorUpdater, boundary_tags, create_distributed_box
    from wave_fenics_amd.linear_gll import LinearGLLOpt, cfl_time_step

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("WF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    p, n = args.degree, args.size
    L = 0.1
    periodic = tuple(c in args.periodic for c in "xyz")
    part = create_distributed_box(n, p, world, rank, hi=(L, L, L), periodic=periodic)
    updater = VectorUpdater(part, device=dev) if (world > 1 or any(periodic)) else None
    eqn = LinearGLLOpt(part.V, p, 1500.0, 0.5e6, 6e4, updater=updater, tags=boundary_tags(part), device=dev)
    dt, _ = cfl_time_step(part.mesh, p, 1500.0, 0.5e6, CFL=0.25)
    eqn.init()
    run = eqn.rk4 if args.unfused else eqn.rk4_fused

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0.0, args.warmup * dt - 1e-13, dt)
    sync()
    t0 = time.perf_counter()
    run(args.warmup * dt, (args.warmup + args.steps) * dt - 1e-13, dt)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    ok = bool(torch.isfinite(eqn.u_n).all())
    if rank == 0:
        print(json.dumps({"metric": "RK4 time step, P4 hex box, full loop (4 x [ghost fwd, K, boundary, ghost rev, vector algebra])",
                          "ms_per_rk4_step": el / args.steps * 1e3, "dof_stages_per_s": 4.0 * part.size_global * args.steps / el,
                          "n_gpus": world, "global_dofs": part.size_global, "cells_per_gpu": part.mesh.ncells,
                          "fused": not args.unfused, "periodic": args.periodic,
                          "exchange": updater.transport if updater is not None else None, "finite": ok, "scaling": "weak"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
