"""Helpers for the multi-process tests (test infrastructure)."""
import os
import socket

import numpy as np


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TorchIndexKernels:
    """Test double for the libwavehip pack/unpack kernels so that the exchange
    logic (partition, index lists, all_to_all sizes) can run on CPU tensors under
    gloo.  Never used by the product path."""

    @staticmethod
    def gather(idx, src, out):
        out.copy_(src[idx.long()])

    @staticmethod
    def scatter_set(idx, src, out):
        out[idx.long()] = src

    @staticmethod
    def scatter_add(idx, src, out):
        out.index_add_(0, idx.long(), src)


def local_to_global(part):
    """Global lattice index of every local lattice point."""
    p = part.degree
    NX, NY, NZ = part.V.lattice
    gn = [part.procs[a] * part.n_local[a] for a in range(3)]
    per = getattr(part, "periodic", (False, False, False))
    G = [p * gn[a] + (0 if per[a] else 1) for a in range(3)]     # periodic axes: the upper plane is the lower one
    K, J, I = np.meshgrid(np.arange(NZ), np.arange(NY), np.arange(NX), indexing="ij")
    Ig = (I + p * part.n_local[0] * part.coords[0]) % G[0]
    Jg = (J + p * part.n_local[1] * part.coords[1]) % G[1]
    Kg = (K + p * part.n_local[2] * part.coords[2]) % G[2]
    return (Ig + G[0] * (Jg + G[1] * Kg)).reshape(-1)


def init_pg(rank, world, port, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
