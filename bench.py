#!/usr/bin/env python3
"""bench.py -- dofs/sec of the stiffness-operator apply on a P4 hex box mesh.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched under torch.distributed.run, one rank per GPU (RCCL).  W untimed
warm-up steps, then EXACTLY K timed steps bracketed by barrier +
torch.cuda.synchronize(); MAX over ranks; rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over the resident synthetic mesh:
    y += K x   (stiffness apply, the metric's operator)   then
    kv = y / m (lumped-mass-inverse apply, BASELINE.json configs[1])
N = 1 : BASELINE.json configs[1]: P4, 54^3 cells, 10 218 313 dofs.
N > 1 : weak scaling, 54^3 cells per GPU on a Cartesian partition of the box,
        forward ghost update of x before and reverse (add) update of y after the
        local apply (common/LinearGLL.hpp:164-176) through VectorUpdater.
value = global owned dofs / (seconds per step)  -- the reference's definition
        index_map->size_local()/t (demo/gpu_operator/main.cpp:171) summed over ranks.
Inputs are resident in HBM before the timed region starts."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(sample_n: int, p: int, reps: int = 5):
    """The reference CPU operator (dense skernel, common/operators.hpp:113-133,
    183-200) restated in oracle/wave_oracle.c, built with the reference's flags,
    one thread per host core over a static cell partition with private y
    (BASELINE.md section 3).  kind = "port"."""
    import concurrent.futures as cf

    import numpy as np
    from oracle import wave_oracle as o
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    om = o.create_box(sample_n, p)
    K = o.StiffnessOperator(om, p, {"c0": 1500.0}, fast=True)
    X = o.dof_coordinates(om)
    x = np.sin(2 * np.pi * X[:, 0])
    nthreads = max(1, min(cores, om.ncells))
    bounds = np.linspace(0, om.ncells, nthreads + 1).astype(int)
    ys = [np.zeros(om.ndofs) for _ in range(nthreads)]

    def work(i):
        K(x, ys[i], cells=(int(bounds[i]), int(bounds[i + 1])))

    times = []
    with cf.ThreadPoolExecutor(nthreads) as ex:
        for r in range(reps + 1):
            for y in ys:
                y[:] = 0.0
            t0 = time.perf_counter()
            list(ex.map(work, range(nthreads)))
            times.append(time.perf_counter() - t0)     # the applies only; the reduction of the private y's is not timed
    t = float(np.median(times[1:]))
    return {
        "value": om.ndofs / t, "unit": "dofs/s", "cores": nthreads, "kind": "port",
        "sample": f"P{p} {sample_n}^3-cell box ({om.ndofs} dofs), dense reference skernel, "
                  f"-Ofast -march=native, {reps} timed applies, median {t:.3f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=54, help="cells per edge per GPU (reference flag --size)")
    ap.add_argument("--degree", type=int, default=4, help="element degree (reference flag --degree)")
    ap.add_argument("--generic", action="store_true", help="use the arbitrary-dofmap kernel instead of the box kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=-1, help="tuning: compiled cross-section of the one-thread-per-column "
                    "marching kernel (0..2); default: the library's choice")
    ap.add_argument("--block", default="", help="tuning: column cross-section bx,by of the k-split marching kernel")
    ap.add_argument("--cpu-sample", type=int, default=24)
    ap.add_argument("--settle-steps", type=int, default=1000, help="untimed steps run before the W warmup steps so that the "
                    "timed region sees the GPU at its sustained operating point: the first ~60 launches after idle run "
                    "up to 25 %% slower while board power ramps from 300 W to 1.2 kW (profiles/r03_power_ramp.md); the "
                    "reference's RK4 loop runs thousands of steps.  0 = time from idle; the from-idle figure is reported "
                    "beside the sustained one either way")
    ap.add_argument("--periodic", default="", help="axes (e.g. xyz) whose opposite faces are identified: every rank "
                    "then has ghost planes on those axes and exchanges them over RCCL -- with one rank, with itself "
                    "(rehearses the multi-GPU exchange + overlap path on one GPU; not the headline configuration)")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON record of rank 0.  Libraries that print on
    # file descriptor 1 (RCCL writes a five-line version banner there when a communicator is created,
    # once per rank) are sent to stderr for the duration of the run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import wave_fenics_amd as w
    from wave_fenics_amd import la

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # WF_BENCH_BACKEND=gloo is a rehearsal hook for a one-GPU box (ranks share the
    # card, halo staged through the host); the driver's multi-GPU run uses nccl = RCCL.
    backend = os.environ.get("WF_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    p, n = args.degree, args.size
    updater = None
    periodic = tuple(c in args.periodic for c in "xyz")
    if world == 1 and not any(periodic):
        mesh = w.create_box(n)
        V = w.create_functionspace(mesh, p, build_dofmap=args.generic)
        V.structured = not args.generic
        owned_global = V.ndofs
        workload = f"P{p} hex box {n}^3 cells, {owned_global} dofs, stiffness + lumped-mass-inverse apply"
        parallelism = "single"
    else:
        from wave_fenics_amd.distributed import create_distributed_box, VectorUpdater
        part = create_distributed_box(n, p, world, rank, periodic=periodic, build_dofmap=args.generic)
        mesh, V = part.mesh, part.V
        V.structured = not args.generic
        # ghost exchange: the C ABI's RCCL updater (grouped ncclSend/ncclRecv per neighbour);
        # WF_UPDATER=torch selects torch.distributed's all_to_all_single on the same index lists
        transport = os.environ.get("WF_UPDATER", "native" if backend == "nccl" else "torch")
        updater, err = None, ""
        try:
            updater = VectorUpdater(part, device=dev, transport=transport)
        except Exception as e:   # e.g. librccl not loadable by the C ABI: agree on it across ranks below
            err = f"{type(e).__name__}: {e}"
        if transport == "native" and world > 1:
            # every rank must take the same transport: if the native RCCL updater failed anywhere, all
            # ranks use torch.distributed's RCCL all_to_all on the same index lists (still no host staging)
            ok = torch.tensor([0 if updater is None else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if rank == 0 or err:
                    print(f"# rank {rank}: native RCCL updater unavailable ({err or 'failed on another rank'}); "
                          "using torch.distributed transport", file=sys.stderr, flush=True)
                if updater is not None:
                    updater.close() if hasattr(updater, "close") else None
                updater = VectorUpdater(part, device=dev, transport="torch")
        elif updater is None:
            raise RuntimeError(err)
        owned_global = part.size_global
        workload = (f"P{p} hex box, {part.procs[0]}x{part.procs[1]}x{part.procs[2]} partition, {n}^3 cells per GPU, "
                    f"{owned_global} dofs, ghost fwd + stiffness + ghost rev(add) + lumped-mass-inverse apply")
        parallelism = (f"dd{world} ({part.procs[0]}x{part.procs[1]}x{part.procs[2]}), exchange={updater.transport}"
                       + (f", periodic={args.periodic}" if any(periodic) else ""))

    tuning = None
    if args.variant >= 0 or args.block:
        tuning = {"variant": args.variant}
        if args.block:
            bx, by = (int(v) for v in args.block.split(","))
            tuning["block"] = (bx, by, 1)
    K = w.StiffnessOperator(V, p, {"c0": 1500.0}, tuning=tuning)
    M = w.MassOperatorLumped(V, p)
    N = V.ndofs
    # synthetic input x = sin(2 pi X) at the dof coordinate (demo/gpu_operator/main.cpp:81), built on device
    NX, NY, NZ = V.lattice
    pts, _, _ = w.tabulate_gll(p)
    ncx = (NX - 1) // p
    xs = np.concatenate([(np.arange(ncx)[:, None] + pts[None, :p]).reshape(-1), [float(ncx)]]) / ncx
    x0 = float(mesh.lo[0]); x1 = float(mesh.hi[0])
    xs = x0 + (x1 - x0) * xs
    x = torch.sin(2 * np.pi * torch.from_numpy(xs).to(dev)).repeat(NY * NZ).contiguous()
    y = torch.zeros(N, dtype=torch.float64, device=dev)
    kv = torch.zeros(N, dtype=torch.float64, device=dev)
    m = torch.zeros(N, dtype=torch.float64, device=dev)
    M(torch.ones(N, dtype=torch.float64, device=dev), m)
    if updater is not None:
        updater.scatter_rev(m)
        updater.scatter_fwd(m)

    split = False
    if updater is not None:
        # cells that read no ghost value run while the halo of x is in flight
        if args.generic:
            split = K.set_ghost_dofs(updater.h_ghost_pos)      # any dofmap: work items whose dof tile holds a ghost position
        else:
            split = K.set_ghost_faces(*[bool(v) for v in part.owned_lo])
    from wave_fenics_amd.distributed import overlapped_apply
    if rank == 0 and updater is not None:
        print(f"# ghost exchange: {updater.transport}, overlap split: {split}", file=sys.stderr, flush=True)

    def step(ev=None):
        if split:
            # interior cells on this stream; halo exchange + interface cells beside them on a second stream
            if ev is not None:
                ev[0].record()
            overlapped_apply(K, updater, x, y)
            if ev is not None:
                ev[1].record()
        else:
            if updater is not None:
                updater.scatter_fwd(x)
            if ev is not None:
                ev[0].record()
            K(x, y)
            if ev is not None:
                ev[1].record()
            if updater is not None:
                updater.scatter_rev(y)
        la.pointwise_div(y, m, kv)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region():
        """W untimed warmup steps, then exactly K timed steps between barrier + synchronize; max over ranks."""
        for _ in range(args.warmup):
            step()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(events[i])
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # average duration of the dominant kernel (stiffness) from HIP events on its stream
        return dt, float(np.mean([a.elapsed_time(b) for a, b in events]))

    from_idle = None
    if args.settle_steps > 0:
        # the same W + K measurement straight after set-up (GPU idle, board power ~300 W): reported, not `value`
        idle_elapsed, idle_kern = timed_region()
        from_idle = {"ms_per_step": idle_elapsed / args.steps * 1e3, "kernel_ms": idle_kern,
                     "value": owned_global / (idle_elapsed / args.steps)}
        for _ in range(args.settle_steps):
            step()
    elapsed, kern_ms = timed_region()
    assert bool(torch.isfinite(kv).all()), "non-finite result"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        alg = K.alg_bytes()
        # N > 1: the event pair brackets the whole overlapped apply (interior + interface + both halos)
        achieved = alg / (kern_ms * 1e-3) / 1e9
        # the box kernel addresses the lattice implicitly and never reads the 4*nd dofmap
        # bytes the contract figure includes: also report the fraction on the bytes it must move
        must_move = alg - (4.0 * (p + 1) ** 3 * mesh.ncells if not args.generic else 0.0)
        # HBM traffic of the kernel from the rocprofv3 PMC passes of tools/evidence.sh (not collected in this run);
        # quoted only when it was measured on THIS build of the library
        traffic, traffic_note = None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and p == 4 and n == 54 and not args.generic and tuning is None:
            try:
                import hashlib
                from wave_fenics_amd import _lib
                with open(_lib.LIB_PATH, "rb") as f:
                    sha = hashlib.sha256(f.read()).hexdigest()[:16]
                with open(tp) as f:
                    tj = json.load(f)
                if tj.get("lib_sha16") == sha:
                    traffic = tj.get("stiffness_hbm_bytes_per_launch")
                    traffic_note = f"static: profiles/traffic.json ({tj.get('tag')}, rocprofv3 --pmc passes on this library build {sha}, not this run)"
                else:
                    traffic_note = f"profiles/traffic.json was measured on library build {tj.get('lib_sha16')}, this is {sha}: not quoted"
            except Exception:
                traffic = None
        out = {
            "metric": "dofs/sec stiffness-operator apply, P4 hex box mesh",
            "value": owned_global / (elapsed / args.steps),
            "unit": "dofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "settle_steps": args.settle_steps, "from_idle": from_idle,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "degree": p, "cells_per_gpu": int(mesh.ncells),
                       "global_dofs": int(owned_global), "parallelism": parallelism,
                       "kernel": "generic" if args.generic else "box"},
            "roofline": {"bound": "hbm", "kernel": "stiffness apply", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_note,
                         "alg_bytes_per_launch": alg, "kernel_ms": kern_ms,
                         "must_move_bytes_per_launch": must_move,
                         "frac_must_move": must_move / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "stiffness_only_dofs_per_s": V.ndofs / (kern_ms * 1e-3)},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample, p)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "dofs/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
