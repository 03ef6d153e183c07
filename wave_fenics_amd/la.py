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
              vn_next=None):
    """Fused stage tail + next stage head (wf_rk4_stage)."""
    has_next = un is not None
    z = 0
    check(lib().wf_rk4_stage(b.numel(), float(bdt), float(adt_next), int(has_next), _ptr(b), _ptr(m), _ptr(vn),
                             _ptr(u_read), _ptr(v_read), _ptr(u), _ptr(v),
                             _ptr(u0) if has_next else z, _ptr(v0) if has_next else z,
                             _ptr(un) if has_next else z, _ptr(vn_next) if has_next else z, _stream(b)))
