"""linalg:: / kernels:: vector operations of the reference on device vectors
(common/cuda/la.hpp:31-138, common/LinearGLL.hpp:15-35)."""
from __future__ import annotations

from ._lib import check, lib
from .operators import _ptr, _stream


def copy(x, y, n: int | None = None):
    """kernels::copy (LinearGLL.hpp:17-21): whole array incl. ghosts unless n given."""
    n = x.numel() if n is None else n
    check(lib().wf_copy(n, _ptr(x), _ptr(y), _stream(x)))


def fill(x, value: float, n: int | None = None):
    n = x.numel() if n is None else n
    check(lib().wf_fill(n, float(value), _ptr(x), _stream(x)))


def axpy(r, alpha: float, x, y, n: int | None = None):
    """kernels::axpy (LinearGLL.hpp:28-33): r = alpha*x + y over the first n
    (= size_local) entries."""
    n = x.numel() if n is None else n
    check(lib().wf_axpy(n, float(alpha), _ptr(x), _ptr(y), _ptr(r), _stream(x)))


def scale(alpha: float, x, n: int | None = None):
    n = x.numel() if n is None else n
    check(lib().wf_scale(n, float(alpha), _ptr(x), _stream(x)))


def pointwise_div(b, m, out, n: int | None = None):
    """out = b / m (LinearGLL.hpp:189-190)."""
    n = b.numel() if n is None else n
    check(lib().wf_pointwise_div(n, _ptr(b), _ptr(m), _ptr(out), _stream(b)))


def pointwise_mult_add(m, x, y, n: int | None = None):
    n = x.numel() if n is None else n
    check(lib().wf_pointwise_mult_add(n, _ptr(m), _ptr(x), _ptr(y), _stream(x)))


def inner_product(x, y, n: int | None = None) -> float:
    """linalg::inner_product (la.hpp:87-103) over size_local entries."""
    import torch
    n = x.numel() if n is None else n
    res = torch.zeros(1, dtype=torch.float64, device=x.device)
    check(lib().wf_dot(n, _ptr(x), _ptr(y), _ptr(res), _stream(x)))
    return float(res.item())


def boundary_apply(idx1, m1, s1: float, idx2, m2, s2: float, v, b):
    """b[idx1] += s1*m1 ; b[idx2] += s2*m2*v[idx2]  (diagonal form of forms.ufl:19-24)."""
    n1 = 0 if idx1 is None else idx1.numel()
    n2 = 0 if idx2 is None else idx2.numel()
    check(lib().wf_boundary_apply(n1, _ptr(idx1) if n1 else 0, _ptr(m1) if n1 else 0, float(s1),
                                  n2, _ptr(idx2) if n2 else 0, _ptr(m2) if n2 else 0, float(s2),
                                  _ptr(v), _ptr(b), _stream(b)))


def rk4_stage(b, m, vn, u_read, v_read, u, v, bdt: float, adt_next: float = 0.0, u0=None, v0=None, un=None,
              vn_next=None, bc=None, s1_next: float = 0.0, s2: float = 0.0):
    """Fused stage tail + next stage head (wf_rk4_stage); with bc (a BoundaryPlan) b is left holding the
    next right-hand side's boundary term instead of zeros (wf_rk4_stage_bc)."""
    has_next = un is not None
    z = 0
    if bc is not None:
        check(lib().wf_rk4_stage_bc(b.numel(), float(bdt), float(adt_next), int(has_next), _ptr(b), _ptr(m), _ptr(vn),
                                    _ptr(u_read), _ptr(v_read), _ptr(u), _ptr(v),
                                    _ptr(u0) if has_next else z, _ptr(v0) if has_next else z,
                                    _ptr(un) if has_next else z, _ptr(vn_next) if has_next else z,
                                    bc._h, float(s1_next), float(s2), _stream(b)))
        return
    check(lib().wf_rk4_stage(b.numel(), float(bdt), float(adt_next), int(has_next), _ptr(b), _ptr(m), _ptr(vn),
                             _ptr(u_read), _ptr(v_read), _ptr(u), _ptr(v),
                             _ptr(u0) if has_next else z, _ptr(v0) if has_next else z,
                             _ptr(un) if has_next else z, _ptr(vn_next) if has_next else z, _stream(b)))


class BoundaryPlan:
    """wf_boundary_create: the two boundary dof sets of the form L (LinearGLL.hpp:113-115, forms.ufl:19-24) as a
    bitmap + coefficient arrays, so that the fused RK4 stage can leave the next right-hand side's boundary
    term in b instead of zeroing it (rk4_stage(..., bc=plan, s1_next=, s2=))."""

    def __init__(self, n: int, idx1, m1, idx2, m2):
        import ctypes

        import numpy as np

        from . import _lib
        i1 = np.ascontiguousarray(idx1, dtype=np.int32)
        i2 = np.ascontiguousarray(idx2, dtype=np.int32)
        a1 = np.ascontiguousarray(m1, dtype=np.float64)
        a2 = np.ascontiguousarray(m2, dtype=np.float64)
        ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if a.size else None
        dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a.size else None
        self._h = ctypes.c_void_p()
        check(lib().wf_boundary_create(int(n), i1.size, ip(i1), dp(a1), i2.size, ip(i2), dp(a2), ctypes.byref(self._h)))

    def apply(self, s1: float, s2: float, v, b):
        """b[idx1] += s1 m1; b[idx2] += s2 m2 v[idx2] (the plain launch)."""
        check(lib().wf_boundary_apply_plan(self._h, float(s1), float(s2), _ptr(v), _ptr(b), _stream(b)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().wf_boundary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cg(x, b, A, kmax: int = 50, rtol: float = 1e-8, updater=None, comm=None):
    """device::cg(x, b, matvec, kmax, rtol) of demo/gpu_cg/CUDA/cg.hpp:38-121 (wf_cg):
    solves A x = b, x holding the initial guess.  A is an operator of this package or a
    callable matvec(v, y) with accumulate semantics (y += A v) on device tensors.
    Returns (iterations, relative residual ||r|| / ||r0||)."""
    import ctypes

    import torch

    from . import _lib
    d = _lib.CGDesc()
    d.n = x.numel()
    keep = None
    if hasattr(A, "_h"):
        d.op = A._h
    else:
        dev, n = x.device, x.numel()

        def tramp(user, pv, py, stream):
            try:
                # wrap the raw pointers as tensors (no copy) and run the callable on wf_cg's stream
                v = _tensor_from_ptr(pv, n, dev)
                y = _tensor_from_ptr(py, n, dev)
                # torch.cuda.ExternalStream(0) is NOT the null stream on ROCm (it wraps a bogus handle)
                st = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
                with torch.cuda.stream(st):
                    A(v, y)
                return 0
            except Exception:   # exceptions must not cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        keep = _lib.MATVEC_FN(tramp)
        d.matvec = keep
    if updater is not None:
        if getattr(updater, "transport", None) != "native":
            raise _lib.WavehipError("cg on a partitioned mesh needs the native (RCCL) VectorUpdater")
        d.updater = updater._h
        comm = comm if comm is not None else updater.comm
    if comm is not None:
        d.comm = comm._h
    d.kmax, d.rtol = int(kmax), float(rtol)
    its, res = ctypes.c_int(0), ctypes.c_double(0.0)
    check(lib().wf_cg(ctypes.byref(d), _ptr(x), _ptr(b), ctypes.byref(its), ctypes.byref(res), _stream(x)))
    return int(its.value), float(res.value)


def _tensor_from_ptr(ptr: int, n: int, device):
    """A float64 tensor view of device memory owned by libwavehip (plumbing for callbacks)."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)
