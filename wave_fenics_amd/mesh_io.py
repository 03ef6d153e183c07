"""Mesh and facet-tag input: the XDMF/HDF5 files the reference's CPU demo reads
(demo/cpu_planar3d/main.cpp:39-45: XDMFFile("../mesh.xdmf").read_mesh(..., "planar3d"),
read_meshtags(mesh, "planar3d_boundaries")), and what DOLFINx derives from them for an
arbitrary conforming hexahedral mesh: the degree-P function space and the tagged boundary
dof sets with their collocated facet masses.

The file I/O is wf_mesh_* of the C ABI (csrc/mesh_io.cpp, HDF5 bound at run time).  The
reference's mesh.xdmf is not in its repository, so this path is pinned by round trips and
by parity with the box meshes, not by a reference file ("parity unpinned")."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_double, c_int32, c_int64, c_void_p
from dataclasses import dataclass

import numpy as np

from ._lib import check, lib
from .box import BoxMesh, FunctionSpace, IndexMap


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double))


def _ip(a):
    return a.ctypes.data_as(POINTER(c_int32))


@dataclass
class MeshTags:
    """mesh::MeshTags<int32> of dimension 2 as read from the file: the four vertices of
    every tagged facet (tensor order) and its value."""
    facet_vertices: np.ndarray   # [nfacets][4] int32
    values: np.ndarray           # [nfacets] int32


def read_mesh(xdmf_path: str, name: str, tags_name: str | None = None):
    """XDMFFile.read_mesh(name) [+ read_meshtags(tags_name)] -> BoxMesh-like mesh (n = None:
    not necessarily a box) [and MeshTags]."""
    h = c_void_p()
    check(lib().wf_mesh_open(xdmf_path.encode(), name.encode(), ctypes.byref(h)))
    try:
        nv, nc = c_int64(), c_int64()
        check(lib().wf_mesh_sizes(h, ctypes.byref(nv), ctypes.byref(nc)))
        x = np.zeros((nv.value, 3))
        cells = np.zeros((nc.value, 8), dtype=np.int32)
        check(lib().wf_mesh_read(h, _dp(x), _ip(cells)))
        mesh = BoxMesh(None, x, cells)
        if tags_name is None:
            return mesh
        nf = c_int64()
        check(lib().wf_mesh_tags_size(h, tags_name.encode(), ctypes.byref(nf)))
        fv = np.zeros((nf.value, 4), dtype=np.int32)
        val = np.zeros(nf.value, dtype=np.int32)
        check(lib().wf_mesh_read_tags(h, tags_name.encode(), _ip(fv), _ip(val)))
        return mesh, MeshTags(fv, val)
    finally:
        lib().wf_mesh_close(h)


def write_mesh(xdmf_path: str, name: str, mesh, tags_name: str | None = None, tags: MeshTags | None = None):
    """Writes <xdmf_path> and the .h5 beside it in the layout DOLFINx' XDMFFile uses."""
    x = np.ascontiguousarray(mesh.x, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    if tags is None:
        check(lib().wf_mesh_write(xdmf_path.encode(), name.encode(), x.shape[0], _dp(x), cells.shape[0], _ip(cells),
                                  None, 0, None, None))
        return
    fv = np.ascontiguousarray(tags.facet_vertices, dtype=np.int32)
    val = np.ascontiguousarray(tags.values, dtype=np.int32)
    check(lib().wf_mesh_write(xdmf_path.encode(), name.encode(), x.shape[0], _dp(x), cells.shape[0], _ip(cells),
                              tags_name.encode(), fv.shape[0], _ip(fv), _ip(val)))


# ---------------------------------------------------------------------------
# what fem::create_functionspace / the form compiler derive from the mesh
# ---------------------------------------------------------------------------
def _q1_shape(X):
    """Trilinear shape functions at reference points X [nq][3] -> [nq][8], vertex v = a + 2b + 4c."""
    out = np.ones((X.shape[0], 8))
    for v in range(8):
        for d in range(3):
            out[:, v] *= X[:, d] if (v >> d) & 1 else 1.0 - X[:, d]
    return out


def create_functionspace(mesh, degree: int) -> FunctionSpace:
    """fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped)) for an
    arbitrary conforming hexahedral mesh: the dofs of a cell sit at the images of the GLL
    nodes under the cell's trilinear map, and two element-local dofs are the same global
    dof exactly when they sit at the same point -- so the dofmap follows from the
    coordinates (quantised to 1e-9 of the smallest edge).  Tensor-ordered, any cell
    orientation.  Dofs are numbered in lexicographic order of (z, y, x)."""
    from .operators import tabulate_gll
    p = int(degree)
    n = p + 1
    pts, _, _ = tabulate_gll(p)
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    X = np.stack([pts[i.reshape(-1)], pts[j.reshape(-1)], pts[k.reshape(-1)]], axis=1)
    N = _q1_shape(X)                                            # [nd][8]
    xc = mesh.x[mesh.geom_dofmap]                               # [c][8][3]
    xd = np.einsum("qv,cvd->cqd", N, xc).reshape(-1, 3)         # [c*nd][3]
    e = np.linalg.norm(xc[:, 1] - xc[:, 0], axis=1).min()
    q = np.round(xd / (1e-9 * e)).astype(np.int64)
    _, first, inv = np.unique(q[:, ::-1], axis=0, return_index=True, return_inverse=True)
    ndofs = int(first.size)
    dm = inv.reshape(mesh.ncells, n ** 3).astype(np.int32)
    V = FunctionSpace(mesh, p, np.ascontiguousarray(dm), IndexMap(ndofs), None, structured=False)
    V.dof_coordinates = xd[first]
    return V


_FACE_VERTS = {(0, 0): (0, 2, 4, 6), (0, 1): (1, 3, 5, 7), (1, 0): (0, 1, 4, 5), (1, 1): (2, 3, 6, 7),
               (2, 0): (0, 1, 2, 3), (2, 1): (4, 5, 6, 7)}


def locate_facets(mesh, tags: MeshTags, value: int):
    """The (cell, axis, side) of every facet carrying `value`: the facet's vertex set is
    matched against the faces of the cells (an exterior facet belongs to one cell)."""
    face_of = {}
    for (axis, side), lv in _FACE_VERTS.items():
        keys = np.sort(mesh.geom_dofmap[:, lv], axis=1)
        for c, kk in enumerate(map(tuple, keys)):
            face_of.setdefault(kk, []).append((c, axis, side))
    out = []
    for fv in np.sort(tags.facet_vertices[tags.values == value], axis=1):
        hits = face_of.get(tuple(fv))
        if not hits:
            raise ValueError("a tagged facet is not a face of any cell")
        out.append(hits[0])
    return out


def facet_lumped_mass(V: FunctionSpace, facets):
    """Collocated facet masses m[i] = sum_facets w_q |dx/ds x dx/dt| (diagonal GLL form of
    inner(g, v) * ds(tag), demo/cpu_planar3d/forms.ufl:19-24) for a list of (cell, axis, side).
    Returns (dof indices int32 ascending, masses)."""
    from .operators import tabulate_gll
    mesh, p = V.mesh, V.degree
    n = p + 1
    pts, w, _ = tabulate_gll(p)
    acc = {}
    bb, aa = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    aa, bb = aa.reshape(-1), bb.reshape(-1)
    for c, axis, side in facets:
        ta, tb = [d for d in range(3) if d != axis]
        X = np.zeros((n * n, 3))
        X[:, axis] = float(side)
        X[:, ta], X[:, tb] = pts[aa], pts[bb]
        xv = mesh.x[mesh.geom_dofmap[c]]                       # [8][3]
        t = []
        for d in (ta, tb):                                     # tangents dx/dX_d of the trilinear map
            g = np.zeros((n * n, 8))
            for v in range(8):
                f = np.ones(n * n)
                for dd in range(3):
                    if dd == d:
                        f = f * (1.0 if (v >> dd) & 1 else -1.0)
                    else:
                        f = f * (X[:, dd] if (v >> dd) & 1 else 1.0 - X[:, dd])
                g[:, v] = f
            t.append(g @ xv)
        ds = np.linalg.norm(np.cross(t[0], t[1]), axis=1) * w[aa] * w[bb]
        loc = np.zeros((n * n, 3), dtype=np.int64)
        loc[:, axis] = side * p
        loc[:, ta], loc[:, tb] = aa, bb
        dofs = V.dofmap[c, loc[:, 0] + n * (loc[:, 1] + n * loc[:, 2])]
        for dof, val in zip(dofs, ds):
            acc[int(dof)] = acc.get(int(dof), 0.0) + float(val)
    idx = np.array(sorted(acc), dtype=np.int32)
    return idx, np.array([acc[int(i)] for i in idx])


def boundary_sets(V: FunctionSpace, tags: MeshTags, values=(1, 2)):
    """The `boundary=` argument of LinearGLLOpt from the file's facet tags
    (tag 1 = Gamma_1 source, tag 2 = Gamma_2 absorbing; common/LinearGLL.hpp:113-115)."""
    return tuple(facet_lumped_mass(V, locate_facets(V.mesh, tags, v)) for v in values)


def cfl_time_step(mesh, degree: int, c0: float, freq: float, CFL: float = 0.5):
    """demo/cpu_planar3d/main.cpp:48-66 (mesh::h = largest vertex distance of a cell)."""
    xc = mesh.x[mesh.geom_dofmap]
    d = np.linalg.norm(xc[:, :, None, :] - xc[:, None, :, :], axis=3)
    h = d.reshape(mesh.ncells, -1).max(axis=1).min()
    dt = CFL * h / (c0 * degree ** 2)
    period = 1.0 / freq
    spp = int(period / dt + 1)
    return period / spp, spp
