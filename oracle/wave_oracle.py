"""
wave_oracle.py -- CPU restatement (numpy + C) of the wave-fenics operator path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product path
(wave_fenics_amd/) never does.

PARITY UNPINNED.  The reference (Excalibur-SLE/wave-fenics) ships no golden
vectors or known-answer tests for this path and cannot be built offline (it
needs DOLFINx, Basix, xtensor and FFCx).  The third-party
arithmetic it calls (Basix GLL quadrature / GLL-warped Lagrange tabulation,
DOLFINx create_box / math::det / math::inv, FFCx facet kernels; no version is
pinned anywhere in the reference, API usage dates it to DOLFINx~0.4/Basix~0.4)
is restated here from the published algorithms:

  * GLL points: roots of P'_{n-1} plus the end points, weights
    2 / (n (n-1) P_{n-1}(x)^2), mapped from [-1,1] to [0,1].
  * GLL-warped Lagrange of degree P on the hexahedron: the tensor product of
    the 1-D Lagrange basis through the P+1 GLL points.
  * Basix/DOLFINx internal orderings (dof order on the hex, quadrature point
    order) cannot be consulted offline.  This restatement uses its own
    lexicographic tensor ordering, local index l = i + n*(j + n*k) with i the
    x-direction; a Basix-ordered run differs only by a permutation of vector
    entries (and rounding).

Each function cites the reference file:line it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------
# C library loading
# --------------------------------------------------------------------------
_LIB = None
_LIB_FAST = None


def _declare(lib):
    i32p = ctypes.POINTER(ctypes.c_int32)
    dp = ctypes.POINTER(ctypes.c_double)
    ci = ctypes.c_int
    lib.oracle_stiffness_apply.argtypes = [ci, ci, ci, ci, i32p, dp, dp, ctypes.c_double, dp, dp]
    lib.oracle_stiffness_apply.restype = None
    lib.oracle_mass_apply.argtypes = [ci, ci, ci, ci, i32p, i32p, dp, dp, dp]
    lib.oracle_mass_apply.restype = None
    lib.oracle_dense_mass_apply.argtypes = [ci, ci, ci, ci, i32p, dp, dp, dp, dp]
    lib.oracle_dense_mass_apply.restype = None
    lib.oracle_geometry.argtypes = [ci, ci, ci, dp, i32p, dp, dp, ci, ci, dp, dp]
    lib.oracle_geometry.restype = None
    lib.oracle_stiffness_apply_sumfact.argtypes = [ci, ci, ci, i32p, dp, dp, ctypes.c_double, dp, dp]
    lib.oracle_stiffness_apply_sumfact.restype = None
    return lib


def build(force: bool = False) -> str:
    """Compile the strict-IEEE oracle library (oracle/Makefile)."""
    so = os.path.join(_HERE, "libwave_oracle.so")
    src = os.path.join(_HERE, "wave_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                               "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _declare(ctypes.CDLL(build()))
    return _LIB


def lib_fast():
    """The same source with the reference's flags (-Ofast -march=native ...),
    compiled on the machine it runs on (march=native must not travel)."""
    global _LIB_FAST
    if _LIB_FAST is None:
        src = os.path.join(_HERE, "wave_oracle.c")
        try:
            with open("/proc/cpuinfo") as f:
                flags = [l for l in f if l.startswith("flags")][0]
        except Exception:
            flags = "unknown"
        with open(src, "rb") as f:
            key = hashlib.sha1(flags.encode() + f.read()).hexdigest()[:16]
        so = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"libwave_oracle_fast_{key}.so")
        if not os.path.exists(so):
            subprocess.check_call(["gcc", "-Ofast", "-march=native", "-mprefer-vector-width=512",
                                   "-fPIC", "-shared", "-o", so, src, "-lm"])
        _LIB_FAST = _declare(ctypes.CDLL(so))
    return _LIB_FAST


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


# --------------------------------------------------------------------------
# a1: tabulation  (common/operators.hpp:13-32)
# --------------------------------------------------------------------------
def gll_points_weights(n: int):
    """n-point Gauss-Lobatto-Legendre rule on [0, 1], points ascending.

    Stands in for basix::quadrature::make_quadrature(gll, interval, m)
    (called at common/operators.hpp:19, common/precomputation.hpp:50).  The
    reference's qdegree map {2:3,3:4,4:6,5:8,6:10,7:12,...}
    (operators.hpp:63-72) selects the rule with n = P+1 points per direction.
    """
    if n < 2:
        raise ValueError("GLL needs n >= 2")
    N = n - 1
    # Chebyshev-Gauss-Lobatto initial guess, Newton on (1-x^2) P'_N(x)
    x = -np.cos(np.pi * np.arange(n) / N)
    for _ in range(100):
        P = np.zeros((n, n))
        P[0] = 1.0
        P[1] = x
        for k in range(2, n):
            P[k] = ((2 * k - 1) * x * P[k - 1] - (k - 1) * P[k - 2]) / k
        dx = (x * P[N] - P[N - 1]) / (n * P[N])
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    P = np.zeros((n, n))
    P[0] = 1.0
    P[1] = x
    for k in range(2, n):
        P[k] = ((2 * k - 1) * x * P[k - 1] - (k - 1) * P[k - 2]) / k
    w = 2.0 / (N * n * P[N] ** 2)
    x[0], x[-1] = -1.0, 1.0
    # symmetrise
    x = 0.5 * (x - x[::-1])
    w = 0.5 * (w + w[::-1])
    return 0.5 * (x + 1.0), 0.5 * w


def gauss_legendre_points_weights(m: int):
    """m-point Gauss-Legendre (Gauss-Jacobi alpha=beta=0) rule on [0, 1].
    Stands in for make_quadrature(gauss_jacobi, ...) (common/precompute.hpp:183-184,
    demo/gpu_operator/main.cpp:96-99)."""
    x, w = np.polynomial.legendre.leggauss(m)
    return 0.5 * (x + 1.0), 0.5 * w


def lagrange_1d(nodes: np.ndarray, pts: np.ndarray):
    """Values and first derivatives of the Lagrange basis through `nodes`
    evaluated at `pts`:  phi[q, a] = l_a(pts[q]), dphi[q, a] = l_a'(pts[q])."""
    nodes = np.asarray(nodes, dtype=np.float64)
    pts = np.asarray(pts, dtype=np.float64)
    n = len(nodes)
    phi = np.ones((len(pts), n))
    dphi = np.zeros((len(pts), n))
    for a in range(n):
        denom = 1.0
        for b in range(n):
            if b != a:
                denom *= nodes[a] - nodes[b]
        for q, xq in enumerate(pts):
            num = 1.0
            for b in range(n):
                if b != a:
                    num *= xq - nodes[b]
            phi[q, a] = num / denom
            s = 0.0
            for c in range(n):
                if c == a:
                    continue
                t = 1.0
                for b in range(n):
                    if b != a and b != c:
                        t *= xq - nodes[b]
                s += t
            dphi[q, a] = s / denom
    return phi, dphi


def clamp101(a: np.ndarray) -> np.ndarray:
    """xt::filtration(t, xt::isclose(t, v)) = v for v in (-1, 0, 1), in that
    order (common/operators.hpp:27-29, common/precomputation.hpp:56-58,105-107).
    xtensor's isclose defaults: rtol 1e-5, atol 1e-8 (same as numpy)."""
    a = np.array(a, dtype=np.float64, copy=True)
    a[np.isclose(a, -1.0)] = -1.0
    a[np.isclose(a, 0.0)] = 0.0
    a[np.isclose(a, 1.0)] = 1.0
    return a


def tabulate_1d_gll(p: int):
    """1-D collocated tables for degree p: (points, weights, phi, D) with
    D[q, a] = l_a'(xi_q); clamped like the reference clamps its dense table."""
    pts, wts = gll_points_weights(p + 1)
    phi, D = lagrange_1d(pts, pts)
    return pts, wts, clamp101(phi), clamp101(D)


def tabulate_basis_and_permutation(p: int, q: int | None = None):
    """common/operators.hpp:13-32.  Returns (perm, table) with
    table[4][nq][nd] (0 = values, 1..3 = d/dx, d/dy, d/dz), clamped.

    q is the reference's Basix quadrature *degree*; only the values of the
    qdegree map (operators.hpp:63-72) are meaningful and all of them select
    the (p+1)-point GLL rule, so it is accepted and checked only.
    perm is the identity: the tables are already in tensor order."""
    qmap = {2: 3, 3: 4, 4: 6, 5: 8, 6: 10, 7: 12, 8: 14, 9: 16, 10: 18}
    if q is not None and p in qmap and q != qmap[p]:
        raise ValueError("quadrature degree outside the reference's qdegree map")
    n = p + 1
    pts, wts = gll_points_weights(n)
    phi1, d1 = lagrange_1d(pts, pts)
    nd = n ** 3
    table = np.zeros((4, nd, nd))
    # local index l = i + n*(j + n*k); tensor product, x index fastest
    # kron order: z (slowest) (x) y (x) x (fastest)
    table[0] = np.kron(phi1, np.kron(phi1, phi1))
    table[1] = np.kron(phi1, np.kron(phi1, d1))
    table[2] = np.kron(phi1, np.kron(d1, phi1))
    table[3] = np.kron(d1, np.kron(phi1, phi1))
    table = clamp101(table)
    perm = np.arange(nd, dtype=np.int32)
    return perm, table


def quadrature_weights_hex(p: int):
    pts, w = gll_points_weights(p + 1)
    W = np.einsum("k,j,i->kji", w, w, w).reshape(-1)
    n = p + 1
    X = np.zeros((n ** 3, 3))
    kk, jj, ii = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    X[:, 0] = pts[ii.reshape(-1)]
    X[:, 1] = pts[jj.reshape(-1)]
    X[:, 2] = pts[kk.reshape(-1)]
    return X, W


# --------------------------------------------------------------------------
# Box mesh (stand-in for dolfinx::mesh::create_box; demo/gpu_operator/main.cpp:60-63)
# --------------------------------------------------------------------------
@dataclass
class BoxMesh:
    n: tuple            # cells per direction (nx, ny, nz)
    p: int              # element degree
    x: np.ndarray       # vertices [nv][3]
    geom_dofmap: np.ndarray   # [ncells][8] int32, vertex v = a + 2b + 4c
    dofmap: np.ndarray  # [ncells][nd] int32, tensor order l = i + n(j + n k)
    ndofs: int
    lattice: tuple      # dof lattice (NX, NY, NZ)
    facet_tags: dict = field(default_factory=dict)
    periodic: tuple = (False, False, False)

    @property
    def ncells(self):
        return self.geom_dofmap.shape[0]


def create_box(n, p: int, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0), perturb: float = 0.0,
               seed: int = 42) -> BoxMesh:
    """Unit-cube style hexahedral box mesh with its own lexicographic numbering
    (vertices and dofs, x fastest).  perturb > 0 displaces interior vertices by
    perturb * h * U(-1, 1) per coordinate (numpy default_rng(seed)), which makes
    all nine entries of G non-zero (SURVEY 8d geometry variant B)."""
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = (int(v) for v in n)
    vx = np.linspace(lo[0], hi[0], nx + 1)
    vy = np.linspace(lo[1], hi[1], ny + 1)
    vz = np.linspace(lo[2], hi[2], nz + 1)
    Z, Y, X = np.meshgrid(vz, vy, vx, indexing="ij")
    x = np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], axis=1).copy()
    if perturb > 0.0:
        rng = np.random.default_rng(seed)
        h = np.array([(hi[0] - lo[0]) / nx, (hi[1] - lo[1]) / ny, (hi[2] - lo[2]) / nz])
        d = rng.uniform(-1.0, 1.0, size=x.shape) * (perturb * h)
        iz, iy, ix = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
        interior = ((ix > 0) & (ix < nx) & (iy > 0) & (iy < ny) & (iz > 0) & (iz < nz)).reshape(-1)
        x[interior] += d[interior]
    cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cx, cy, cz = cx.reshape(-1), cy.reshape(-1), cz.reshape(-1)
    gd = np.zeros((nx * ny * nz, 8), dtype=np.int32)
    for v in range(8):
        a, b, c = v & 1, (v >> 1) & 1, (v >> 2) & 1
        gd[:, v] = (cx + a) + (nx + 1) * ((cy + b) + (ny + 1) * (cz + c))
    nn = p + 1
    NX, NY, NZ = p * nx + 1, p * ny + 1, p * nz + 1
    k, j, i = np.meshgrid(np.arange(nn), np.arange(nn), np.arange(nn), indexing="ij")
    i, j, k = i.reshape(-1), j.reshape(-1), k.reshape(-1)
    dm = ((p * cx[:, None] + i[None, :])
          + NX * ((p * cy[:, None] + j[None, :]) + NY * (p * cz[:, None] + k[None, :])))
    return BoxMesh((nx, ny, nz), p, x, gd, dm.astype(np.int32), NX * NY * NZ, (NX, NY, NZ))


def make_periodic(mesh: BoxMesh, periodic) -> np.ndarray:
    """Identify the upper face of every axis in `periodic` with the lower one: the
    dofmap is renumbered onto the reduced lattice (upper-plane dofs take the number
    of their lower-plane image), exterior facets on those axes disappear
    (box_facets).  Test infrastructure for the ghost-exchange parity tests: a
    periodic operator is what local apply + self/neighbour exchange must equal.
    Returns the map old lattice index -> new dof number."""
    NX, NY, NZ = mesh.lattice
    per = tuple(bool(v) for v in periodic)
    K, J, I = np.meshgrid(np.arange(NZ), np.arange(NY), np.arange(NX), indexing="ij")
    RX, RY, RZ = (NX - 1 if per[0] else NX), (NY - 1 if per[1] else NY), (NZ - 1 if per[2] else NZ)
    new = ((I % RX) + RX * ((J % RY) + RY * (K % RZ))).reshape(-1)
    mesh.dofmap = new[mesh.dofmap].astype(np.int32)
    mesh.ndofs = RX * RY * RZ
    mesh.periodic = per
    return new


def cmap_tabulate(X: np.ndarray):
    """Q1 coordinate-element tabulation (geometry.cmap().tabulate(1, points),
    common/precomputation.hpp:55): phi[q][v], dphi[3][q][v], v = a + 2b + 4c."""
    nq = X.shape[0]
    phi = np.zeros((nq, 8))
    dphi = np.zeros((3, nq, 8))
    for v in range(8):
        bits = (v & 1, (v >> 1) & 1, (v >> 2) & 1)
        f = [X[:, d] if bits[d] else 1.0 - X[:, d] for d in range(3)]
        g = [np.ones(nq) if bits[d] else -np.ones(nq) for d in range(3)]
        phi[:, v] = f[0] * f[1] * f[2]
        dphi[0, :, v] = g[0] * f[1] * f[2]
        dphi[1, :, v] = f[0] * g[1] * f[2]
        dphi[2, :, v] = f[0] * f[1] * g[2]
    return phi, dphi


def dof_coordinates(mesh: BoxMesh) -> np.ndarray:
    """Physical coordinates of every dof (push-forward of the GLL nodes)."""
    X, _ = quadrature_weights_hex(mesh.p)
    phi, _ = cmap_tabulate(X)
    xc = mesh.x[mesh.geom_dofmap]              # [c][8][3]
    xd = np.einsum("qv,cvd->cqd", phi, xc)     # [c][nd][3]
    out = np.zeros((mesh.ndofs, 3))
    out[mesh.dofmap.reshape(-1)] = xd.reshape(-1, 3)
    return out


# --------------------------------------------------------------------------
# a2: geometry  (common/precomputation.hpp:18-110)
# --------------------------------------------------------------------------
def precompute_geometric_data(mesh: BoxMesh, p: int | None = None):
    """Returns (G[ncells][nq][3][3], detJ[ncells][nq]) exactly as
    common/precomputation.hpp:69-107: detJ = |det J| * w_q,
    G = (J^-1 * detJ) . J^-T, then clamp of G to -1/0/1."""
    p = mesh.p if p is None else p
    X, W = quadrature_weights_hex(p)
    _, dphi = cmap_tabulate(X)
    dphi = clamp101(dphi)                       # precomputation.hpp:56-58
    nq = X.shape[0]
    G = np.zeros((mesh.ncells, nq, 3, 3))
    detJ = np.zeros((mesh.ncells, nq))
    xv = np.ascontiguousarray(mesh.x)
    gd = np.ascontiguousarray(mesh.geom_dofmap)
    dphi = np.ascontiguousarray(dphi)
    W = np.ascontiguousarray(W)
    lib().oracle_geometry(mesh.ncells, nq, 8, _dp(xv), _ip(gd), _dp(dphi), _dp(W), 1, 1,
                          _dp(G), _dp(detJ))
    return G, detJ


def compute_detJ_generic(mesh: BoxMesh, X: np.ndarray, W: np.ndarray):
    """The generic path used by the GPU mass operators: det(J) * w WITHOUT fabs
    (common/precompute.hpp:49-116, common/cuda/mass.hpp:35-39,
    common/cuda/spectral_mass.hpp:58-64)."""
    _, dphi = cmap_tabulate(X)
    nq = X.shape[0]
    detJ = np.zeros((mesh.ncells, nq))
    xv = np.ascontiguousarray(mesh.x)
    gd = np.ascontiguousarray(mesh.geom_dofmap)
    dphi = np.ascontiguousarray(dphi)
    W = np.ascontiguousarray(W)
    lib().oracle_geometry(mesh.ncells, nq, 8, _dp(xv), _ip(gd), _dp(dphi), _dp(W), 0, 0,
                          None, _dp(detJ))
    return detJ


# --------------------------------------------------------------------------
# a3/a4: stiffness  (common/operators.hpp:113-133, 137-201)
# --------------------------------------------------------------------------
class StiffnessOperator:
    """common/operators.hpp:137-201.  op(x, y): y += K x (y is not zeroed)."""

    def __init__(self, mesh: BoxMesh, bdegree: int, params: dict | None = None, fast: bool = False):
        self.mesh = mesh
        self.ndofs = (bdegree + 1) ** 3
        self.G, self.detJ = precompute_geometric_data(mesh, bdegree)
        self.perm, table = tabulate_basis_and_permutation(bdegree)
        self.dphi = np.ascontiguousarray(table[1:4])
        # operators.hpp:114 hard-codes c0 = 1500 and ignores params; the only
        # caller passes the same value (demo/cpu_planar3d/main.cpp:25).
        self.c0 = 1500.0 if params is None else float(params.get("c0", 1500.0))
        self._lib = lib_fast() if fast else lib()

    def __call__(self, x: np.ndarray, y: np.ndarray, cells=None):
        c0, c1 = (0, self.mesh.ncells) if cells is None else cells
        nq = self.detJ.shape[1]
        self._lib.oracle_stiffness_apply(c0, c1, self.ndofs, nq, _ip(self.mesh.dofmap),
                                         _dp(self.G), _dp(self.dphi), self.c0, _dp(x), _dp(y))


def stiffness_apply_sumfact(mesh: BoxMesh, G: np.ndarray, c0: float, x: np.ndarray, y: np.ndarray,
                            fast: bool = False):
    """Sum-factorised CPU variant (BASELINE.md 'Baseline B'), informative."""
    _, _, _, D = tabulate_1d_gll(mesh.p)
    D = np.ascontiguousarray(D)
    L = lib_fast() if fast else lib()
    L.oracle_stiffness_apply_sumfact(0, mesh.ncells, mesh.p + 1, _ip(mesh.dofmap), _dp(G), _dp(D),
                                     c0, _dp(x), _dp(y))


# --------------------------------------------------------------------------
# a5: lumped mass (common/operators.hpp:36-40, 44-109)
# --------------------------------------------------------------------------
class MassOperatorCPU:
    """common/operators.hpp:44-109.  op(x, y): y += M_lumped x."""

    def __init__(self, mesh: BoxMesh, bdegree: int):
        self.mesh = mesh
        self.ndofs = (bdegree + 1) ** 3
        self.G, self.detJ = precompute_geometric_data(mesh, bdegree)
        self.perm, table = tabulate_basis_and_permutation(bdegree)
        self.phi = np.ascontiguousarray(table[0])

    def __call__(self, x: np.ndarray, y: np.ndarray):
        nq = self.detJ.shape[1]
        lib().oracle_mass_apply(0, self.mesh.ncells, self.ndofs, nq, _ip(self.mesh.dofmap),
                                _ip(self.perm), _dp(self.detJ), _dp(x), _dp(y))


MassOperator = MassOperatorCPU   # the name common/LinearGLL.hpp:63,105 uses


def dense_mass_apply(mesh: BoxMesh, phi: np.ndarray, detJ: np.ndarray, x: np.ndarray, y: np.ndarray):
    """common/cuda/mass.hpp:76-95 + mass_kernel.cu:5-37: y += Phi^T D Phi x."""
    nq, nd = phi.shape
    phi = np.ascontiguousarray(phi)
    lib().oracle_dense_mass_apply(0, mesh.ncells, nd, nq, _ip(mesh.dofmap), _dp(phi), _dp(detJ),
                                  _dp(x), _dp(y))


def tabulate_mass_tables(p: int, variant: str, quad: str, qdegree: int):
    """Tables for the GPU MassOperator demos.
      variant 'gll'        : GLL-warped Lagrange (demo/gpu_operator_monolithic/main.cpp:69-71)
      variant 'equispaced' : equispaced Lagrange (demo/gpu_operator/main.cpp:66-68)
      quad 'gll'           : m-point GLL rule; the demo's qdegree = degree+1 (>1)
                             selects the (p+1)-point rule (monolithic main.cpp:94-96)
      quad 'gauss_jacobi'  : Gauss-Legendre with ceil((qdegree+1)/2) points
                             (demo/gpu_operator/main.cpp:96-99, qdegree = 2*degree)
    Returns 1-D (pts, wts, phi1[nq1][n1]) and the dense 3-D phi[nq][nd], X, W."""
    n = p + 1
    nodes = gll_points_weights(n)[0] if variant == "gll" else np.linspace(0.0, 1.0, n)
    if quad == "gll":
        m = n if p > 1 else 2
        pts, wts = gll_points_weights(m)
    else:
        m = (qdegree + 2) // 2
        pts, wts = gauss_legendre_points_weights(m)
    phi1, _ = lagrange_1d(nodes, pts)
    phi = np.kron(phi1, np.kron(phi1, phi1))
    W = np.einsum("k,j,i->kji", wts, wts, wts).reshape(-1)
    kk, jj, ii = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    X = np.stack([pts[ii.reshape(-1)], pts[jj.reshape(-1)], pts[kk.reshape(-1)]], axis=1)
    return pts, wts, phi1, phi, X, W


# --------------------------------------------------------------------------
# a7: boundary operator (demo/cpu_planar3d/forms.ufl:19-24)
# --------------------------------------------------------------------------
def box_facets(mesh: BoxMesh):
    """Exterior facets of the box as (cell, local_face) with tags
    1 = face x == lo (Gamma_1, Neumann source), 2 = every other face
    (Gamma_2, absorbing) -- SURVEY 8d, cfg1.  local_face = 2*axis + side."""
    nx, ny, nz = mesh.n
    out = []
    cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cid = (cx + nx * (cy + ny * cz))
    for axis, (cc, nn) in enumerate(((cx, nx), (cy, ny), (cz, nz))):
        if mesh.periodic[axis]:
            continue
        for side in (0, 1):
            sel = cid[cc == (0 if side == 0 else nn - 1)].reshape(-1)
            tag = 1 if (axis == 0 and side == 0) else 2
            out.append((sel.astype(np.int32), 2 * axis + side, tag))
    return out


def facet_lumped_mass(mesh: BoxMesh, tag: int) -> np.ndarray:
    """m_Gamma[i] = sum over facets with `tag` of w_q |J_facet| at the facet's
    GLL points, which are collocated with the facet dofs, so the facet mass
    matrix is diagonal.  Restates the FFCx kernel for
    inner(g, v) * ds(tag, metadata={'quadrature_rule': 'GLL', 'quadrature_degree': 6})
    (demo/cpu_planar3d/forms.ufl:19-24); the generated forms.c is git-ignored in
    the reference, so this is parity unpinned."""
    p = mesh.p
    n = p + 1
    pts, w = gll_points_weights(n)
    m = np.zeros(mesh.ndofs)
    for cells, lf, t in box_facets(mesh):
        if t != tag:
            continue
        axis, side = lf // 2, lf % 2
        ta, tb = [d for d in range(3) if d != axis]
        # facet points: reference coords with X[axis] = side
        bb, aa = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        aa, bb = aa.reshape(-1), bb.reshape(-1)
        X = np.zeros((n * n, 3))
        X[:, axis] = float(side)
        X[:, ta] = pts[aa]
        X[:, tb] = pts[bb]
        _, dphi = cmap_tabulate(X)
        xc = mesh.x[mesh.geom_dofmap[cells]]                  # [f][8][3]
        J = np.einsum("fvi,jqv->fqij", xc, dphi)              # J[i][j] = dx_i/dX_j
        t1 = J[:, :, :, ta]
        t2 = J[:, :, :, tb]
        nrm = np.linalg.norm(np.cross(t1, t2), axis=2)        # [f][q]
        wq = (w[aa] * w[bb])[None, :] * nrm
        idx = np.zeros(3, dtype=object)
        loc = np.zeros((n * n, 3), dtype=np.int64)
        loc[:, axis] = side * p
        loc[:, ta] = aa
        loc[:, tb] = bb
        l = loc[:, 0] + n * (loc[:, 1] + n * loc[:, 2])
        dofs = mesh.dofmap[cells][:, l]                        # [f][n*n]
        np.add.at(m, dofs.reshape(-1), wq.reshape(-1))
    return m


# --------------------------------------------------------------------------
# a6/a15: LinearGLLOpt  (common/LinearGLL.hpp:37-288)
# --------------------------------------------------------------------------
class LinearGLLOpt:
    """Single-process restatement of common/LinearGLL.hpp (no ghosts: the
    scatter_fwd / scatter_rev calls at :110,127,164,167,176 are identities on
    one rank).  The boundary form L (forms.ufl:19-24) is applied in its
    collocated diagonal form (facet_lumped_mass)."""

    def __init__(self, mesh: BoxMesh, degreeOfBasis: int, speedOfSound: float,
                 sourceFrequency: float, pressureAmplitude: float):
        self.mesh = mesh
        self.k_ = degreeOfBasis
        self.c0_ = speedOfSound
        self.freq0_ = sourceFrequency
        self.p0_ = pressureAmplitude
        self.w0_ = 2.0 * np.pi * self.freq0_
        self.T_ = 1.0 / self.freq0_
        self.alpha_ = 4.0
        N = mesh.ndofs
        # LinearGLL.hpp:102-110  m = M * 1
        u = np.ones(N)
        self.mass_op = MassOperator(mesh, self.k_)
        self.m = np.zeros(N)
        self.mass_op(u, self.m)
        # LinearGLL.hpp:113-115 boundary form
        self.mG1 = facet_lumped_mass(mesh, 1)
        self.mG2 = facet_lumped_mass(mesh, 2)
        # LinearGLL.hpp:120-127
        self.stiff_op = StiffnessOperator(mesh, self.k_, {"c0": self.c0_})
        self.b = np.zeros(N)
        self.u_n = np.zeros(N)
        self.v_n = np.zeros(N)
        self.g = 0.0

    def init(self):
        self.u_n[:] = 0.0
        self.v_n[:] = 0.0

    def f0(self, t, u, v, result):
        result[:] = v                                       # LinearGLL.hpp:141-144

    def f1(self, t, u, v, result):
        # LinearGLL.hpp:151-192
        if t < self.T_ * self.alpha_:
            window = 0.5 * (1.0 - np.cos(self.freq0_ * np.pi * t / self.alpha_))
        else:
            window = 1.0
        self.g = window * self.p0_ * self.w0_ / self.c0_ * np.cos(self.w0_ * t)
        self.u_n[:] = u
        self.v_n[:] = v
        self.b[:] = 0.0
        self.stiff_op(self.u_n, self.b)
        # fem::assemble_vector(_b, *L): L = c0^2 * ( g v ds(1) - (1/c0) v_n v ds(2) )
        self.b += self.c0_ ** 2 * (self.g * self.mG1 - (1.0 / self.c0_) * self.mG2 * self.v_n)
        result[:] = self.b / self.m

    def rk4(self, startTime: float, finalTime: float, timeStep: float, max_steps: int | None = None):
        # LinearGLL.hpp:198-287
        t, tf, dt = startTime, finalTime, timeStep
        step = 0
        u_ = self.u_n.copy()
        v_ = self.v_n.copy()
        un = np.zeros_like(u_)
        vn = np.zeros_like(u_)
        u0 = np.zeros_like(u_)
        v0 = np.zeros_like(u_)
        ku = u_.copy()
        kv = v_.copy()
        a_runge = [0.0, 0.5, 0.5, 1.0]
        b_runge = [1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0]
        c_runge = [0.0, 0.5, 0.5, 1.0]
        while t < tf:
            dt = min(dt, tf - t)
            u0[:] = u_
            v0[:] = v_
            for i in range(4):
                un[:] = u0
                vn[:] = v0
                un[:] = ku * (dt * a_runge[i]) + un     # kernels::axpy: vx*alpha + vy
                vn[:] = kv * (dt * a_runge[i]) + vn
                tn = t + c_runge[i] * dt
                self.f0(tn, un, vn, ku)
                self.f1(tn, un, vn, kv)
                u_[:] = ku * (dt * b_runge[i]) + u_
                v_[:] = kv * (dt * b_runge[i]) + v_
            t += dt
            step += 1
            if max_steps is not None and step >= max_steps:
                break
        self.u_n[:] = u_
        self.v_n[:] = v_
        return t, step


def cfl_time_step(mesh: BoxMesh, degree: int, c0: float, freq: float, CFL: float = 0.5):
    """demo/cpu_planar3d/main.cpp:48-66: dt = CFL*h/(c0*P^2), rounded so that an
    integer number of steps fits one source period.  h = minimum cell diameter
    (mesh::h: largest vertex-to-vertex distance of a cell)."""
    xc = mesh.x[mesh.geom_dofmap]
    d = np.linalg.norm(xc[:, :, None, :] - xc[:, None, :, :], axis=3)
    h = d.reshape(mesh.ncells, -1).max(axis=1).min()
    dt = CFL * h / (c0 * degree ** 2)
    period = 1.0 / freq
    stepPerPeriod = int(period / dt + 1)
    return period / stepPerPeriod, stepPerPeriod
