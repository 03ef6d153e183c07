"""Host-side box mesh and function-space descriptors.

Stand-ins for the DOLFINx objects the reference passes to its operators
(mesh::create_box + fem::create_functionspace, demo/gpu_operator/main.cpp:60-72):
plain numpy arrays with this engine's lexicographic numbering
(vertex (a,b,c) -> a + (nx+1)(b + (ny+1)c); dof (I,J,K) -> I + NX(J + NY K);
element-local tensor index l = i + n(j + n k))."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class IndexMap:
    """dolfinx::common::IndexMap as far as the operators use it: owned dofs
    first, ghosts after (operators use size_local + num_ghosts as array length)."""
    size_local: int
    num_ghosts: int = 0
    size_global: int | None = None

    def __post_init__(self):
        if self.size_global is None:
            self.size_global = self.size_local


@dataclass
class BoxMesh:
    n: tuple                   # cells per direction
    x: np.ndarray              # vertices [nv][3] float64
    geom_dofmap: np.ndarray    # [ncells][8] int32
    lo: tuple = (0.0, 0.0, 0.0)
    hi: tuple = (1.0, 1.0, 1.0)

    @property
    def ncells(self) -> int:
        return int(self.geom_dofmap.shape[0])


@dataclass
class FunctionSpace:
    mesh: BoxMesh
    degree: int
    dofmap: np.ndarray         # [ncells][nd] int32, tensor order
    index_map: IndexMap
    lattice: tuple             # (NX, NY, NZ)
    structured: bool = True    # dofmap is the implicit lexicographic box numbering

    @property
    def ndofs(self) -> int:
        return self.index_map.size_local + self.index_map.num_ghosts


def lattice_numbering(V: "FunctionSpace") -> np.ndarray:
    """Setup-time renumbering option (wf_lattice_numbering): new index of every dof of V in a numbering that follows
    the lattice columns the marching kernels walk.  For spaces whose own numbering scatters a cell's dofs over
    memory (a uniformly random numbering costs 2.8x in the stiffness apply)."""
    from ._lib import check, lib
    dm = np.ascontiguousarray(V.dofmap, dtype=np.int32)
    out = np.empty(V.ndofs, dtype=np.int32)
    import ctypes
    ip = ctypes.POINTER(ctypes.c_int32)
    check(lib().wf_lattice_numbering(V.degree, dm.shape[0], V.ndofs, dm.ctypes.data_as(ip), out.ctypes.data_as(ip)))
    return out


def renumber(V: "FunctionSpace", new_of_old: np.ndarray) -> "FunctionSpace":
    """The same space with dof d renamed new_of_old[d]; vectors move as x_new[new_of_old] = x_old."""
    return FunctionSpace(V.mesh, V.degree, np.ascontiguousarray(new_of_old[V.dofmap].astype(np.int32)), V.index_map, V.lattice,
                         structured=False)


def create_box(n, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0), perturb: float = 0.0, seed: int = 42) -> BoxMesh:
    """mesh::create_box(comm, {lo, hi}, {nx, ny, nz}, hexahedron).  perturb > 0
    displaces interior vertices by perturb*h*U(-1,1) (numpy default_rng(seed))."""
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
    gd = np.empty((nx * ny * nz, 8), dtype=np.int32)
    for v in range(8):
        a, b, c = v & 1, (v >> 1) & 1, (v >> 2) & 1
        gd[:, v] = (cx + a) + (nx + 1) * ((cy + b) + (ny + 1) * (cz + c))
    return BoxMesh((nx, ny, nz), np.ascontiguousarray(x), gd, tuple(lo), tuple(hi))


def create_functionspace(mesh: BoxMesh, degree: int, build_dofmap: bool = True) -> FunctionSpace:
    """fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped))."""
    nx, ny, nz = mesh.n
    p = int(degree)
    nn = p + 1
    NX, NY, NZ = p * nx + 1, p * ny + 1, p * nz + 1
    if build_dofmap:
        cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        cx, cy, cz = cx.reshape(-1), cy.reshape(-1), cz.reshape(-1)
        k, j, i = np.meshgrid(np.arange(nn), np.arange(nn), np.arange(nn), indexing="ij")
        i, j, k = i.reshape(-1), j.reshape(-1), k.reshape(-1)
        base = (p * cx + NX * (p * cy + NY * (p * cz))).astype(np.int64)
        off = (i + NX * (j + NY * k)).astype(np.int64)
        dm = (base[:, None] + off[None, :]).astype(np.int32)
    else:
        dm = np.empty((0, nn ** 3), dtype=np.int32)
    return FunctionSpace(mesh, p, dm, IndexMap(NX * NY * NZ), (NX, NY, NZ))
