"""Diagnostic: what of the RCCL path can be exercised on a one-GPU box.
Two ranks on one device are refused by RCCL ("Duplicate GPU detected"), so only
the single-rank group can run: init, barrier, all_reduce, all_to_all_single with
unequal (and zero) splits -- the calls bench.py and VectorUpdater make."""
import os

import torch
import torch.distributed as dist

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", device_id=dev)
    dist.barrier()
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = 5 if world == 1 else 2 * world
    a = torch.arange(n, dtype=torch.float64, device=dev) + 10 * rank
    b = torch.zeros(n, dtype=torch.float64, device=dev)
    splits = [n] if world == 1 else [2] * world
    w = dist.all_to_all_single(b, a, splits, splits, async_op=True)
    w.wait()
    e_in = torch.zeros(1, dtype=torch.float64, device=dev)[:0]
    e_out = torch.zeros(1, dtype=torch.float64, device=dev)[:0]
    w = dist.all_to_all_single(e_out, e_in, [0] * world, [0] * world, async_op=True)   # a rank with no neighbours
    w.wait()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        w = dist.all_to_all_single(b, a, splits, splits, async_op=True)
        w.wait()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(rank, "ok", float(t), b.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(rank, "FAILED", type(e).__name__, str(e)[:400], flush=True)
