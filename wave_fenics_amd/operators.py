"""Host-side mirror of the reference's operator classes on top of libwavehip.

Same names, argument meaning and error behaviour as the reference:
  StiffnessOperator(V, bdegree, params)      common/operators.hpp:137-201
  MassOperatorLumped(V, bdegree)             common/operators.hpp:44-109 (MassOperatorCPU)
  SpectralMassOperator(V, bdegree)           common/cuda/spectral_mass.hpp:24-100
  MassOperator(V, element, quad_type, qd)    common/cuda/mass.hpp:18-107
  gather / scatter / transform1              common/cuda/scatter.hpp, transform.hpp
`op(x, y)` and `op.apply(x, y)` both compute y += A x on device vectors
(torch.float64 CUDA tensors, or anything with data_ptr()).  Failures raise
WavehipError (the reference throws std::runtime_error)."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_double, c_int32, c_void_p

import numpy as np

from . import _lib
from ._lib import OpDesc, OpInfo, check, lib
from .box import FunctionSpace


def _ptr(t) -> int:
    if isinstance(t, int):
        return t
    return int(t.data_ptr())


def _stream(t=None) -> int:
    """The HIP stream the call is ordered on: torch's current stream for the
    tensor's device (plumbing only)."""
    try:
        import torch
        if t is not None and hasattr(t, "is_cuda") and t.is_cuda:
            return int(torch.cuda.current_stream(t.device).cuda_stream)
    except ImportError:
        pass
    return 0


def _check_vec(t, n: int, name: str):
    if hasattr(t, "dtype"):
        import torch
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise _lib.WavehipError(f"{name}: expected a contiguous float64 device vector")
        if t.numel() < n:
            raise _lib.WavehipError(f"{name}: vector has {t.numel()} entries, operator needs {n}")


def _dp(a):
    return None if a is None else a.ctypes.data_as(POINTER(c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(POINTER(c_int32))


# ---------------------------------------------------------------------------
# setup functions (a1, a2)
# ---------------------------------------------------------------------------
def tabulate_gll(p: int):
    """1-D GLL points, weights and collocation derivative matrix D[q][a]."""
    n = p + 1
    pts, wts, D = np.zeros(n), np.zeros(n), np.zeros((n, n))
    check(lib().wf_tabulate_gll(p, _dp(pts), _dp(wts), _dp(D)))
    return pts, wts, D


def tabulate_dense(p: int):
    """tabulate_basis_and_permutation (common/operators.hpp:13-32): returns
    (perm, table[4][nq][nd]); perm is the identity in this engine's ordering."""
    nd = (p + 1) ** 3
    table = np.zeros((4, nd, nd))
    check(lib().wf_tabulate_dense(p, _dp(table)))
    return np.arange(nd, dtype=np.int32), table


_QUAD = {"gll": _lib.WF_QUAD_GLL, "gauss_jacobi": _lib.WF_QUAD_GAUSS_JACOBI}
_VARIANT = {"gll_warped": _lib.WF_VARIANT_GLL_WARPED, "gll": _lib.WF_VARIANT_GLL_WARPED, "equispaced": _lib.WF_VARIANT_EQUISPACED}


def quadrature_1d(quad: str, degree: int):
    """basix::quadrature::make_quadrature(quad, interval, degree) on [0,1]: (points, weights).
    "gauss_jacobi" has (degree+2)//2 points, "gll" (degree+4)//2 (precompute.hpp:183-184,
    operators.hpp:19)."""
    n = ctypes.c_int(0)
    pts, wts = np.zeros(_lib.WF_MAX_QUAD_POINTS), np.zeros(_lib.WF_MAX_QUAD_POINTS)
    check(lib().wf_quadrature_1d(_QUAD[quad], int(degree), ctypes.byref(n), _dp(pts), _dp(wts)))
    return pts[: n.value].copy(), wts[: n.value].copy()


def tabulate_1d(p: int, points, derivative: int = 0, variant: str = "gll_warped"):
    """tabulate_1d(p, q, derivative) of common/precompute.hpp:179-189 at given points:
    table[q][a] = l_a(x_q) or l_a'(x_q) for the degree-p Lagrange basis of `variant`."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    out = np.zeros((pts.size, p + 1))
    check(lib().wf_tabulate_1d(p, _VARIANT[variant], pts.size, _dp(pts), int(derivative), _dp(out)))
    return out


def compute_geometry_rule(mesh, points1, weights1, use_fabs: bool = False, clamp: bool = False, want_G: bool = False):
    """compute_jacobian / _determinant / compute_geometrical_factor of
    common/precompute.hpp:49-176 at the tensor rule points1^3: (G or None, detJ*w)."""
    pts = np.ascontiguousarray(points1, dtype=np.float64)
    wts = np.ascontiguousarray(weights1, dtype=np.float64)
    nq = pts.size ** 3
    x = np.ascontiguousarray(mesh.x, dtype=np.float64)
    gd = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    G = np.zeros((mesh.ncells, nq, 3, 3)) if want_G else None
    detJ = np.zeros((mesh.ncells, nq))
    check(lib().wf_geometry_hex_rule(mesh.ncells, x.shape[0], _dp(x), _ip(gd), pts.size, _dp(pts), _dp(wts),
                                     int(use_fabs), int(clamp), _dp(G), _dp(detJ)))
    return G, detJ


def precompute_geometric_data(mesh, p: int, use_fabs: bool = True, clamp: bool = True, want_G: bool = True):
    """precompute_geometric_data (common/precomputation.hpp:18-110) on the device;
    returns host arrays (G[ncells][nq][3][3], detJ[ncells][nq])."""
    nq = (p + 1) ** 3
    x = np.ascontiguousarray(mesh.x, dtype=np.float64)
    gd = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    G = np.zeros((mesh.ncells, nq, 3, 3)) if want_G else None
    detJ = np.zeros((mesh.ncells, nq))
    check(lib().wf_geometry_hex(p, mesh.ncells, x.shape[0], _dp(x), _ip(gd), int(use_fabs), int(clamp),
                                _dp(G), _dp(detJ)))
    return G, detJ


# ---------------------------------------------------------------------------
# operator handles
# ---------------------------------------------------------------------------
def make_tuning(tuning) -> "_lib.Tuning | None":
    """wf_tuning from a dict (kernel=, variant=, lz=, lz0=, block=(bx, by, bz), keep_cell_order=, orient=) or None.
    `kernel` is a WF_KERNEL_FORCE_* value or one of "batch", "box_block", "mass_any", "elementwise", "march"."""
    if tuning is None:
        return None
    if isinstance(tuning, _lib.Tuning):
        return tuning
    names = {"auto": 0, "batch": _lib.WF_KERNEL_FORCE_BATCH, "box_block": _lib.WF_KERNEL_FORCE_BOX_BLOCK,
             "mass_any": _lib.WF_KERNEL_FORCE_MASS_ANY, "elementwise": _lib.WF_KERNEL_FORCE_ELEMENTWISE,
             "march": _lib.WF_KERNEL_FORCE_MARCH}
    t = _lib.Tuning()
    k = tuning.get("kernel", 0)
    t.kernel = names[k] if isinstance(k, str) else int(k)
    t.variant = int(tuning.get("variant", -1)) + 1       # C ABI: compiled cross-section + 1, 0 = default
    t.lz = int(tuning.get("lz", 0))
    t.lz0 = int(tuning.get("lz0", 0))
    t.bx, t.by, t.bz = (int(v) for v in tuning.get("block", (0, 0, 0)))
    t.keep_cell_order = int(bool(tuning.get("keep_cell_order", False)))
    t.orient = int(tuning.get("orient", 0))
    return t


KERNEL_NAMES = {0: "none", 1: "march_box", 2: "march_idx", 3: "batch_unique", 4: "box_block", 5: "diagonal",
                6: "mass_dense_any", 7: "dense_simplex", 8: "elementwise"}


class _Operator:
    _kind = None

    def __init__(self):
        self._h = c_void_p()

    def _create(self, desc: OpDesc, keep=(), tuning=None):
        self._keep = keep
        t = make_tuning(tuning)
        if t is not None:
            desc.tuning = ctypes.pointer(t)
        check(lib().wf_op_create(ctypes.byref(desc), ctypes.byref(self._h)))
        self._keep = ()
        self._info()

    def _create_box(self, kind: int, p: int, mesh, c0: float, flags: int, tuning=None):
        nx, ny, nz = mesh.n
        x = np.ascontiguousarray(mesh.x, dtype=np.float64)
        t = make_tuning(tuning)
        check(lib().wf_op_create_box_tuned(kind, p, nx, ny, nz, _dp(x), float(c0), flags,
                                           ctypes.byref(t) if t is not None else None, ctypes.byref(self._h)))
        self._info()

    @property
    def kernel(self) -> str:
        """Name of the kernel wf_op_apply launches (wf_op_info_t.kernel)."""
        return KERNEL_NAMES.get(self.info.kernel, "?")

    def _info(self):
        info = OpInfo()
        check(lib().wf_op_info(self._h, ctypes.byref(info)))
        self.info = info

    # introspection as common/cuda/mass.hpp:68-71
    def num_quads(self): return self.info.num_quads
    def num_cells(self): return self.info.num_cells
    def num_dofs(self): return self.info.num_dofs_cell
    def flops(self): return self.info.flops
    def alg_bytes(self): return self.info.alg_bytes

    def apply(self, x, y, stream: int | None = None):
        """y += A x (accumulate; the caller zeroes y, LinearGLL.hpp:173)."""
        _check_vec(x, self.info.ndofs, "x")
        _check_vec(y, self.info.ndofs, "y")
        s = _stream(x) if stream is None else stream
        check(lib().wf_op_apply(self._h, _ptr(x), _ptr(y), s))

    __call__ = apply

    def set_ghost_faces(self, gx: bool, gy: bool, gz: bool) -> bool:
        """Declare which lower lattice planes are ghost planes (domain
        decomposition, box operators).  Returns False when this operator cannot be
        split this way: the caller then uses set_ghost_dofs or the unsplit sequence."""
        rc = lib().wf_op_set_ghost_faces(self._h, int(gx), int(gy), int(gz))
        if rc == -2:
            return False
        check(rc)
        self._info()
        return True

    def set_ghost_dofs(self, ghost_positions) -> bool:
        """Interior / interface split from the ghost positions of the local array (the
        list the VectorUpdater unpacks into): works for every marching operator, box or
        arbitrary dofmap.  Returns False for operators that run a batch kernel."""
        g = np.ascontiguousarray(ghost_positions, dtype=np.int32)
        rc = lib().wf_op_set_ghost_dofs(self._h, _ip(g) if g.size else None, int(g.size))
        if rc == -2:
            return False
        check(rc)
        self._info()
        return True

    def part_fraction(self, part: int) -> float:
        """Share of the operator's work items in `part` (after set_ghost_faces)."""
        tot = self.info.items_interior + self.info.items_interface
        if tot == 0:
            return 1.0
        return (self.info.items_interior if part == 1 else self.info.items_interface) / tot

    def apply_part(self, x, y, part: int, stream: int | None = None):
        """y += A_part x, part in (WF_PART_INTERIOR, WF_PART_INTERFACE):
        interior cells read no ghost value of x."""
        _check_vec(x, self.info.ndofs, "x")
        _check_vec(y, self.info.ndofs, "y")
        s = _stream(x) if stream is None else stream
        check(lib().wf_op_apply_part(self._h, _ptr(x), _ptr(y), part, s))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().wf_op_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _base_desc(V: FunctionSpace, kind: int, degree: int, perm=None):
    d = OpDesc()
    d.kind = kind
    d.degree = degree
    d.ncells = V.mesh.ncells
    d.ndofs = V.ndofs
    keep = []
    dm = np.ascontiguousarray(V.dofmap, dtype=np.int32)
    if dm.shape != (V.mesh.ncells, (degree + 1) ** 3):
        raise _lib.WavehipError("dofmap shape does not match ncells x (degree+1)^3")
    keep.append(dm)
    d.h_dofmap = _ip(dm)
    if perm is not None:
        pm = np.ascontiguousarray(perm, dtype=np.int32)
        keep.append(pm)
        d.h_perm = _ip(pm)
    return d, keep


def _attach_mesh(d: OpDesc, V: FunctionSpace, keep: list):
    x = np.ascontiguousarray(V.mesh.x, dtype=np.float64)
    gd = np.ascontiguousarray(V.mesh.geom_dofmap, dtype=np.int32)
    keep += [x, gd]
    d.nverts = x.shape[0]
    d.h_xverts = _dp(x)
    d.h_geom_dofmap = _ip(gd)


class StiffnessOperator(_Operator):
    """StiffnessOperator(V, bdegree, params) -- common/operators.hpp:137-201.

    params["c0"] is honoured (the reference hard-codes 1500, operators.hpp:114,
    which is the only value its caller passes).  G=None computes the geometry on
    the device from V.mesh (precomputation.hpp:69-107); a host array in the
    reference layout [ncells][nq][3][3] is used as given.
    structured=None picks the implicit-dofmap box kernel when V says its dofmap
    is the lexicographic box numbering; structured=False forces the generic
    (arbitrary dofmap, atomic scatter) kernel."""

    def __init__(self, V: FunctionSpace, bdegree: int, params: dict | None = None, G=None, perm=None,
                 structured: bool | None = None, flags: int = 0, tuning=None):
        super().__init__()
        c0 = 1500.0 if not params else float(params.get("c0", 1500.0))
        self.c0 = c0
        if structured is None:
            structured = bool(getattr(V, "structured", False)) and G is None and perm is None
        if structured:
            if bdegree != V.degree:
                raise _lib.WavehipError("structured operator: bdegree must equal the space's degree")
            self._create_box(_lib.WF_OP_STIFFNESS, bdegree, V.mesh, c0, flags, tuning)
            return
        d, keep = _base_desc(V, _lib.WF_OP_STIFFNESS, bdegree, perm)
        d.c0 = c0
        d.flags = flags
        if G is not None:
            Gc = np.ascontiguousarray(G, dtype=np.float64)
            if Gc.size != V.mesh.ncells * (bdegree + 1) ** 3 * 9:
                raise _lib.WavehipError("G must be [ncells][nq][3][3]")
            keep.append(Gc)
            d.h_G = _dp(Gc)
        else:
            _attach_mesh(d, V, keep)
        self._create(d, keep, tuning)


class MassOperatorLumped(_Operator):
    """MassOperatorCPU(V, bdegree) -- common/operators.hpp:44-109: the GLL-lumped
    mass y += M x.  detJ = |det J| w (fabs), as precompute_geometric_data."""

    def __init__(self, V: FunctionSpace, bdegree: int, detJ=None, perm=None, structured: bool | None = None,
                 flags: int = 0, tuning=None):
        super().__init__()
        if structured is None:
            structured = bool(getattr(V, "structured", False)) and detJ is None and perm is None
        if structured:
            self._create_box(_lib.WF_OP_MASS_LUMPED, bdegree, V.mesh, 0.0, flags)
            return
        d, keep = _base_desc(V, _lib.WF_OP_MASS_LUMPED, bdegree, perm)
        d.flags = flags
        if detJ is not None:
            Dc = np.ascontiguousarray(detJ, dtype=np.float64)
            keep.append(Dc)
            d.h_detJ = _dp(Dc)
        else:
            _attach_mesh(d, V, keep)
        self._create(d, keep, tuning)


class SpectralMassOperator(MassOperatorLumped):
    """SpectralMassOperator(V, bdegree) -- common/cuda/spectral_mass.hpp:24-100.
    Same lumped mass; detJ = det(J) w WITHOUT fabs (spectral_mass.hpp:58-64 via
    precompute.hpp:102-116).  Degrees outside 2..7 are rejected like the
    reference's qdegree map (spectral_mass.hpp:42-48)."""

    def __init__(self, V: FunctionSpace, bdegree: int, structured: bool | None = None):
        if bdegree < 2 or bdegree > 7:
            raise _lib.WavehipError("SpectralMassOperator: degree must be 2..7")
        super().__init__(V, bdegree, structured=structured, flags=_lib.WF_FLAG_NO_FABS)


class MassOperator(_Operator):
    """MassOperator(V, element, quad_type, qd) -- common/cuda/mass.hpp:18-107:
    y += Phi^T diag(detJ w) Phi x with a tensor-product rule.

    Either the reference's arguments -- the element as (degree, variant in
    {"gll_warped", "equispaced"}), quad in {"gll", "gauss_jacobi"} and the quadrature
    degree qdegree (mass.hpp:20-21; demo/gpu_operator/main.cpp:66-68,96-99 uses
    equispaced + gauss_jacobi of degree 2P) -- from which the 1-D table
    (tabulate_1d) and det J * w at the rule's points (on the device) are built; or
    explicit tables: `phi1` [nq1][P+1] and `detJ` [ncells][nq1^3] (mass.hpp:35-39)."""

    def __init__(self, V: FunctionSpace, degree: int, phi1: np.ndarray | None = None, detJ: np.ndarray | None = None,
                 perm=None, variant: str = "gll_warped", quad: str = "gll", qdegree: int | None = None, tuning=None):
        super().__init__()
        d, keep = _base_desc(V, _lib.WF_OP_MASS_DENSE, degree, perm)
        if phi1 is None:
            if qdegree is None:
                qdegree = degree + 1 if degree > 1 else degree      # gpu_operator_monolithic/main.cpp:95
            pts, wts = quadrature_1d(quad, qdegree)
            phi1 = tabulate_1d(degree, pts, 0, variant)
            self.points1, self.weights1 = pts, wts
        p1 = np.ascontiguousarray(phi1, dtype=np.float64)
        if p1.ndim != 2 or p1.shape[1] != degree + 1:
            raise _lib.WavehipError("phi1 must be [nq1][degree+1]")
        keep.append(p1)
        d.nq1 = p1.shape[0]
        d.h_phi1 = _dp(p1)
        if detJ is not None:
            Dc = np.ascontiguousarray(detJ, dtype=np.float64)
            if Dc.size != V.mesh.ncells * p1.shape[0] ** 3:
                raise _lib.WavehipError("detJ must be [ncells][nq1^3]")
            keep.append(Dc)
            d.h_detJ = _dp(Dc)
        else:
            if not hasattr(self, "points1"):
                raise _lib.WavehipError("MassOperator: explicit phi1 needs detJ")
            _attach_mesh(d, V, keep)
            d.flags |= _lib.WF_FLAG_NO_FABS   # mass.hpp:35-39: det J * w keeps its sign (compute_jacobian_determinant)
            qp, qw = np.ascontiguousarray(self.points1), np.ascontiguousarray(self.weights1)
            keep += [qp, qw]
            d.h_qpts1, d.h_qwts1 = _dp(qp), _dp(qw)
        self._create(d, keep, tuning)


# ---------------------------------------------------------------------------
# free kernels (common/cuda/scatter.hpp:7-14, transform.hpp:7-8)
# ---------------------------------------------------------------------------
def gather(N: int, indices, inp, out, block_size: int = 512, stream: int | None = None):
    """out[i] = in[indices[i]] (block_size accepted for signature parity, unused)."""
    check(lib().wf_gather(N, _ptr(indices), _ptr(inp), _ptr(out), _stream(inp) if stream is None else stream))


def scatter(N: int, indices, inp, out, block_size: int = 512, stream: int | None = None):
    """out[indices[i]] += in[i] (hardware fp64 atomics)."""
    check(lib().wf_scatter_add(N, _ptr(indices), _ptr(inp), _ptr(out), _stream(inp) if stream is None else stream))


def scatter_set(N: int, indices, inp, out, stream: int | None = None):
    check(lib().wf_scatter_set(N, _ptr(indices), _ptr(inp), _ptr(out), _stream(inp) if stream is None else stream))


def transform1(N: int, inp, detJ, out, block_size: int = 512, stream: int | None = None):
    """out[i] = in[i] * detJ[i]."""
    check(lib().wf_transform1(N, _ptr(inp), _ptr(detJ), _ptr(out), _stream(inp) if stream is None else stream))


def tsmm(ncells: int, inp, phi, out, layout: int = 0, stream: int | None = None):
    """out[cell][n] = sum_k in[cell][k] phi[k][n] (wf_tsmm; demo/gpu_tsmm, demo/gpu_operator DGEMMs).
    phi: device tensor [K][N] row-major."""
    K, N = int(phi.shape[0]), int(phi.shape[1])
    check(lib().wf_tsmm(layout, ncells, K, N, _ptr(inp), _ptr(phi), _ptr(out), _stream(inp) if stream is None else stream))
