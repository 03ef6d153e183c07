"""
tet_oracle.py -- CPU restatement of the reference's dense operator on tetrahedra
(BASELINE.json configs[4]: "P4 tet unstructured mesh, non-tensor-product operator
path").  TEST INFRASTRUCTURE ONLY (see wave_oracle.py).

PARITY UNPINNED.  The reference's StiffnessOperator is written for hexahedra only
(`_ndofs = (bdegree+1)^3`, common/operators.hpp:154), but its element kernel
`skernel` (common/operators.hpp:113-133) and cell loop (:183-200) are cell-type
agnostic: dense table dphi[3][nq][nd], G[nq][3][3].  This module feeds that same
dense kernel (oracle_stiffness_apply in wave_oracle.c) with tetrahedral tables:

  * element: Lagrange P_p on the reference tetrahedron (0,0,0),(1,0,0),(0,1,0),
    (0,0,1) with equispaced nodes (i,j,k)/p, i+j+k <= p, ordered k slowest, i fastest
    (Basix would supply a GLL-warped variant; the nodal set only changes the basis
    of the same polynomial space, and Basix is not available offline);
  * quadrature: collapsed Gauss-Jacobi (Stroud conical product) with m points per
    direction, exact to degree 2m-1 -- Basix's `gauss_jacobi` scheme, m = (q+2)/2
    for quadrature degree q (common/precompute.hpp:183-184 uses the same scheme);
  * geometry: G = (J^-1 |det J| w_q) J^-T and the -1/0/1 clamp exactly as
    common/precomputation.hpp:95-107, with the affine tetrahedral Jacobian;
  * mesh: Kuhn split of the box mesh (6 tetrahedra per cube), dofs on the
    (p n + 1)^3 lattice.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np
from scipy.special import roots_jacobi

from . import wave_oracle as wo


def tet_nodes(p: int) -> np.ndarray:
    """Integer node coordinates (i, j, k), i + j + k <= p; k slowest, i fastest."""
    out = []
    for k in range(p + 1):
        for j in range(p + 1 - k):
            for i in range(p + 1 - k - j):
                out.append((i, j, k))
    return np.array(out, dtype=np.int64)


def tet_quadrature(m: int):
    """Collapsed Gauss-Jacobi rule on the reference tetrahedron, m^3 points."""
    def rule(alpha):
        t, w = roots_jacobi(m, alpha, 0.0)
        return 0.5 * (t + 1.0), w / 2.0 ** (alpha + 1)
    x1, w1 = rule(2.0)
    x2, w2 = rule(1.0)
    x3, w3 = rule(0.0)
    X, W = [], []
    for a in range(m):
        for b in range(m):
            for c in range(m):
                x = x1[a]
                y = x2[b] * (1.0 - x)
                z = x3[c] * (1.0 - x) * (1.0 - x2[b])
                X.append((x, y, z))
                W.append(w1[a] * w2[b] * w3[c])
    return np.array(X), np.array(W)


def tabulate_tet(p: int, X: np.ndarray):
    """phi[q][d], dphi[3][q][d] of the equispaced Lagrange basis at points X."""
    nodes = tet_nodes(p)
    xn = nodes / float(p)
    expo = nodes                      # monomials x^a y^b z^c over the same index set

    def monos(P):
        return np.prod(P[:, None, :] ** expo[None, :, :], axis=2)

    def dmonos(P, axis):
        e = expo.copy()
        fac = e[:, axis].astype(float)
        e[:, axis] = np.maximum(e[:, axis] - 1, 0)
        return np.prod(P[:, None, :] ** e[None, :, :], axis=2) * fac[None, :]

    V = monos(xn)                     # [node][mono]
    C = np.linalg.inv(V)              # [mono][basis]
    phi = monos(X) @ C
    dphi = np.stack([dmonos(X, a) @ C for a in range(3)])
    return phi, dphi


@dataclass
class TetMesh:
    n: tuple
    p: int
    x: np.ndarray             # vertices [nv][3]
    geom_dofmap: np.ndarray   # [ncells][4] int32
    dofmap: np.ndarray        # [ncells][nd] int32
    ndofs: int
    lattice: tuple

    @property
    def ncells(self):
        return self.geom_dofmap.shape[0]


def create_kuhn_box(n, p: int, perturb: float = 0.0, seed: int = 42, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0)) -> TetMesh:
    """Box of n cubes per direction, each split into 6 Kuhn tetrahedra
    (v0 = cube corner, v1 = v0 + e_pi0, v2 = v1 + e_pi1, v3 = v2 + e_pi2 for the 6
    permutations pi).  Cell c = 6 * cube + permutation index."""
    hexm = wo.create_box(n, 1, lo=lo, hi=hi, perturb=perturb, seed=seed)
    nx, ny, nz = hexm.n
    NX, NY, NZ = p * nx + 1, p * ny + 1, p * nz + 1
    nodes = tet_nodes(p)
    perms = list(itertools.permutations(range(3)))
    cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cube = np.stack([cx.reshape(-1), cy.reshape(-1), cz.reshape(-1)], axis=1)      # [ncube][3]
    ncube = cube.shape[0]
    gd = np.zeros((ncube, 6, 4), dtype=np.int64)
    dm = np.zeros((ncube, 6, nodes.shape[0]), dtype=np.int64)
    vstride = np.array([1, nx + 1, (nx + 1) * (ny + 1)])
    dstride = np.array([1, NX, NX * NY])
    for ip, pi in enumerate(perms):
        off = np.zeros((4, 3), dtype=np.int64)
        for s in range(3):
            off[s + 1] = off[s]
            off[s + 1, pi[s]] += 1
        for v in range(4):
            gd[:, ip, v] = ((cube + off[v]) * vstride).sum(axis=1)
        # node (i,j,k): lattice offset i*e_pi0 + j*(e_pi0+e_pi1) + k*(1,1,1)
        lat = np.zeros((nodes.shape[0], 3), dtype=np.int64)
        lat[:, pi[0]] = nodes[:, 0] + nodes[:, 1] + nodes[:, 2]
        lat[:, pi[1]] = nodes[:, 1] + nodes[:, 2]
        lat[:, pi[2]] = nodes[:, 2]
        dm[:, ip, :] = ((p * cube[:, None, :] + lat[None, :, :]) * dstride).sum(axis=2)
    return TetMesh((nx, ny, nz), p, hexm.x, gd.reshape(-1, 4).astype(np.int32),
                   dm.reshape(ncube * 6, -1).astype(np.int32), NX * NY * NZ, (NX, NY, NZ))


def tet_geometry(mesh: TetMesh, W: np.ndarray):
    """G[ncells][nq][3][3], detJ[ncells][nq] as common/precomputation.hpp:83-107
    with the (constant) affine Jacobian J[i][j] = (v_{j+1} - v_0)[i]."""
    xc = mesh.x[mesh.geom_dofmap]                          # [c][4][3]
    J = np.stack([xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0], xc[:, 3] - xc[:, 0]], axis=2)   # J[c][i][j]
    det = np.linalg.det(J)
    Ji = np.linalg.inv(J)
    detJ = np.abs(det)[:, None] * W[None, :]               # precomputation.hpp:95
    # G = (J_inv * detJ) . J_inv^T                          # precomputation.hpp:99-100
    G = np.einsum("cik,cq,cjk->cqij", Ji, detJ, Ji)
    return wo.clamp101(G), detJ


class TetStiffnessOperator:
    """StiffnessOperator::operator() (common/operators.hpp:183-200) with
    tetrahedral tables: y += K x through the dense skernel."""

    def __init__(self, mesh: TetMesh, p: int, qdegree: int | None = None, c0: float = 1500.0, fast: bool = False):
        self.mesh = mesh
        q = 2 * p - 2 if qdegree is None else qdegree
        self.m = (q + 2) // 2
        self.X, self.W = tet_quadrature(self.m)
        phi, dphi = tabulate_tet(p, self.X)
        self.phi = phi
        self.dphi = np.ascontiguousarray(wo.clamp101(dphi))      # operators.hpp:27-29
        self.G, self.detJ = tet_geometry(mesh, self.W)
        self.G = np.ascontiguousarray(self.G)
        self.c0 = c0
        self.nd = self.dphi.shape[2]
        self.nq = self.dphi.shape[1]
        self._lib = wo.lib_fast() if fast else wo.lib()

    def __call__(self, x, y, cells=None):
        c0, c1 = (0, self.mesh.ncells) if cells is None else cells
        dm = np.ascontiguousarray(self.mesh.dofmap)
        self._lib.oracle_stiffness_apply(c0, c1, self.nd, self.nq, wo._ip(dm), wo._dp(self.G), wo._dp(self.dphi),
                                         self.c0, wo._dp(x), wo._dp(y))


def dof_coordinates(mesh: TetMesh) -> np.ndarray:
    nodes = tet_nodes(mesh.p) / float(mesh.p)
    xc = mesh.x[mesh.geom_dofmap]
    lam0 = 1.0 - nodes.sum(axis=1)
    X = (lam0[None, :, None] * xc[:, None, 0, :] + nodes[None, :, 0, None] * xc[:, None, 1, :]
         + nodes[None, :, 1, None] * xc[:, None, 2, :] + nodes[None, :, 2, None] * xc[:, None, 3, :])
    out = np.zeros((mesh.ndofs, 3))
    out[mesh.dofmap.reshape(-1)] = X.reshape(-1, 3)
    return out
