"""RCCL communicator of libwavehip (wf_comm_*, include/wavehip.h) for Python hosts.

The reference's MPI communicator (demo/gpu_scatter_mpi/main.cpp:52-62,
VectorUpdater.hpp:67-98) becomes an RCCL communicator created INSIDE the C ABI;
torch.distributed (any backend) is used only to hand the 128-byte unique id from
rank 0 to the other ranks -- the role MPI_Bcast would play in a C++ launcher."""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_void_p

from . import _lib
from ._lib import check, lib
from .operators import _ptr, _stream


class Comm:
    def __init__(self, handle: c_void_p, rank: int, size: int):
        self._h = handle
        self.rank = rank
        self.size = size

    # -- construction ---------------------------------------------------------
    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(_lib.WF_COMM_ID_BYTES)
        check(lib().wf_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def create(cls, uid: bytes, rank: int, size: int) -> "Comm":
        """ncclCommInitRank on the CURRENT device (call torch.cuda.set_device / wf_set_device first)."""
        h = c_void_p()
        check(lib().wf_comm_create(uid, rank, size, ctypes.byref(h)))
        return cls(h, rank, size)

    @classmethod
    def single(cls) -> "Comm":
        """A one-rank communicator (self-neighbour / periodic exchanges on one GPU)."""
        return cls.create(cls.unique_id(), 0, 1)

    @classmethod
    def from_torch_distributed(cls, group=None) -> "Comm":
        """One RCCL rank per torch.distributed rank; the id travels by broadcast_object_list."""
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls.single()
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        # Every rank first proves that it can reach RCCL through the C ABI (ncclGetUniqueId loads the
        # library) and the ranks agree on the outcome, so that a rank that cannot does not leave the
        # others waiting in the collectives below.
        uid, err = None, ""
        try:
            uid = cls.unique_id()
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        import torch
        flag_dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        ok = torch.tensor([1 if uid is not None else 0], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise _lib.WavehipError("RCCL is not reachable through libwavehip on every rank" + (f" (this rank: {err})" if err else ""))
        obj = [uid if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(obj, src=src, group=group)
        return cls.create(obj[0], rank, size)

    @classmethod
    def from_file(cls, path: str, rank: int, size: int, timeout_s: float = 120.0) -> "Comm":
        h = c_void_p()
        check(lib().wf_comm_create_from_file(path.encode(), rank, size, float(timeout_s), ctypes.byref(h)))
        return cls(h, rank, size)

    # -- collectives ------------------------------------------------------------
    def rccl_version(self) -> int:
        v = c_int()
        check(lib().wf_comm_info(self._h, None, None, ctypes.byref(v)))
        return int(v.value)

    def allreduce(self, t, op: str = "sum", out=None):
        """MPI_Allreduce(MPI_SUM / MPI_MAX) on a float64 device tensor (in place by default)."""
        out = t if out is None else out
        check(lib().wf_comm_allreduce(self._h, _lib.WF_SUM if op == "sum" else _lib.WF_MAX, t.numel(), _ptr(t), _ptr(out),
                                      _stream(t)))
        return out

    def barrier(self, stream: int = 0):
        check(lib().wf_comm_barrier(self._h, stream))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().wf_comm_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
