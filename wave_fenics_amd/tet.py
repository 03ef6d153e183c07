"""Tetrahedral (non-tensor-product) operator path: host-side tabulation, Kuhn box
mesh and the dense MFMA stiffness operator (csrc/stiffness_dense.hip).

The reference has no tetrahedral operator class (its StiffnessOperator fixes
`_ndofs = (bdegree+1)^3`, common/operators.hpp:154); its element kernel and cell
loop (operators.hpp:113-133,183-200) are cell-agnostic and are what this path
implements for BASELINE.json configs[4].  Tabulation stands in for Basix
(Lagrange P_p on the tetrahedron, equispaced nodes; collapsed Gauss-Jacobi rule
with m = p points per direction, i.e. quadrature degree 2p-2 under Basix's
gauss_jacobi scheme, common/precompute.hpp:183-184)."""
from __future__ import annotations

import ctypes
import itertools
from ctypes import c_void_p
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import DenseDesc, check, lib
from .box import create_box
from .operators import _Operator, _dp, _ip


def tet_nodes(p: int) -> np.ndarray:
    """Integer node coordinates (i, j, k) with i + j + k <= p (k slowest, i fastest)."""
    return np.array([(i, j, k) for k in range(p + 1) for j in range(p + 1 - k) for i in range(p + 1 - k - j)],
                    dtype=np.int64)


def tet_quadrature(m: int):
    """Collapsed Gauss-Jacobi (Stroud) rule on the reference tetrahedron: m^3 points, exact to degree 2m-1."""
    from scipy.special import roots_jacobi
    rules = []
    for alpha in (2.0, 1.0, 0.0):
        t, w = roots_jacobi(m, alpha, 0.0)
        rules.append((0.5 * (t + 1.0), w / 2.0 ** (alpha + 1.0)))
    (x1, w1), (x2, w2), (x3, w3) = rules
    a, b, c = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    a, b, c = a.reshape(-1), b.reshape(-1), c.reshape(-1)
    X = np.stack([x1[a], x2[b] * (1.0 - x1[a]), x3[c] * (1.0 - x1[a]) * (1.0 - x2[b])], axis=1)
    return X, w1[a] * w2[b] * w3[c]


def tabulate_tet(p: int, X: np.ndarray):
    """phi[q][d] and dphi[3][q][d] of the equispaced Lagrange basis of degree p."""
    e = tet_nodes(p)
    xn = e / float(p)
    V = np.prod(xn[:, None, :] ** e[None, :, :], axis=2)
    C = np.linalg.inv(V)
    phi = np.prod(X[:, None, :] ** e[None, :, :], axis=2) @ C
    dphi = []
    for ax in range(3):
        ed = e.copy()
        fac = ed[:, ax].astype(float)
        ed[:, ax] = np.maximum(ed[:, ax] - 1, 0)
        dphi.append((np.prod(X[:, None, :] ** ed[None, :, :], axis=2) * fac[None, :]) @ C)
    return phi, np.stack(dphi)


def clamp101(a):
    a = np.array(a, dtype=np.float64, copy=True)
    a[np.isclose(a, -1.0)] = -1.0
    a[np.isclose(a, 0.0)] = 0.0
    a[np.isclose(a, 1.0)] = 1.0
    return a


@dataclass
class TetSpace:
    n: tuple
    degree: int
    x: np.ndarray             # vertices [nv][3]
    geom_dofmap: np.ndarray   # [ncells][4] int32
    dofmap: np.ndarray        # [ncells][nd] int32
    ndofs: int
    lattice: tuple

    @property
    def ncells(self) -> int:
        return int(self.geom_dofmap.shape[0])


def create_kuhn_box(n, p: int, perturb: float = 0.0, seed: int = 42, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0)) -> TetSpace:
    """Box of n cubes per direction, each cube split into its 6 Kuhn tetrahedra;
    dofs live on the (p n + 1)^3 lattice (cell = 6 * cube + permutation)."""
    hexm = create_box(n, lo, hi, perturb, seed)
    nx, ny, nz = hexm.n
    NX, NY, NZ = p * nx + 1, p * ny + 1, p * nz + 1
    nodes = tet_nodes(p)
    cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cube = np.stack([cx.reshape(-1), cy.reshape(-1), cz.reshape(-1)], axis=1)
    ncube = cube.shape[0]
    gd = np.zeros((ncube, 6, 4), dtype=np.int64)
    dm = np.zeros((ncube, 6, nodes.shape[0]), dtype=np.int64)
    vstride = np.array([1, nx + 1, (nx + 1) * (ny + 1)])
    dstride = np.array([1, NX, NX * NY])
    for ip, pi in enumerate(itertools.permutations(range(3))):
        off = np.zeros((4, 3), dtype=np.int64)
        for s in range(3):
            off[s + 1] = off[s]
            off[s + 1, pi[s]] += 1
        for v in range(4):
            gd[:, ip, v] = ((cube + off[v]) * vstride).sum(axis=1)
        lat = np.zeros((nodes.shape[0], 3), dtype=np.int64)
        lat[:, pi[0]] = nodes.sum(axis=1)
        lat[:, pi[1]] = nodes[:, 1] + nodes[:, 2]
        lat[:, pi[2]] = nodes[:, 2]
        dm[:, ip, :] = ((p * cube[:, None, :] + lat[None, :, :]) * dstride).sum(axis=2)
    return TetSpace((nx, ny, nz), p, hexm.x, gd.reshape(-1, 4).astype(np.int32),
                    dm.reshape(ncube * 6, -1).astype(np.int32), NX * NY * NZ, (NX, NY, NZ))


class TetStiffnessOperator(_Operator):
    """y += K x on affine tetrahedra through wf_op_create_dense_simplex:
    the dense skernel (common/operators.hpp:113-133) on v_mfma_f64_16x16x4_f64."""

    def __init__(self, V: TetSpace, degree: int, params: dict | None = None, qdegree: int | None = None,
                 flags: int = 0):
        super().__init__()
        c0 = 1500.0 if not params else float(params.get("c0", 1500.0))
        q = 2 * degree - 2 if qdegree is None else qdegree
        m = (q + 2) // 2
        X, W = tet_quadrature(m)
        _, dphi = tabulate_tet(degree, X)
        dphi = np.ascontiguousarray(clamp101(dphi))          # operators.hpp:27-29
        W = np.ascontiguousarray(W)
        d = DenseDesc()
        d.nd, d.nq = dphi.shape[2], dphi.shape[1]
        d.ncells, d.ndofs = V.ncells, V.ndofs
        dm = np.ascontiguousarray(V.dofmap, dtype=np.int32)
        x = np.ascontiguousarray(V.x, dtype=np.float64)
        gd = np.ascontiguousarray(V.geom_dofmap, dtype=np.int32)
        d.h_dofmap, d.h_dphi, d.h_weights = _ip(dm), _dp(dphi), _dp(W)
        d.nverts, d.h_xverts, d.h_geom_dofmap = x.shape[0], _dp(x), _ip(gd)
        d.c0, d.flags = c0, flags
        self._h = c_void_p()
        check(lib().wf_op_create_dense_simplex(ctypes.byref(d), ctypes.byref(self._h)))
        self._info()
