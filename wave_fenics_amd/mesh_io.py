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
# what fem::create_functionspace / the form compiler derive from the mesh: wf_fs_* of the C ABI
# (csrc/function_space.cpp), shared with the C++ host code (include/wavehip_mesh.hpp)
# ---------------------------------------------------------------------------
def create_functionspace(mesh, degree: int) -> FunctionSpace:
    """fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped)) for an
    arbitrary conforming hexahedral mesh whose cells may have ANY local orientation: dofs are
    identified topologically (by mesh entity and canonical position on it, never by comparing
    coordinates) and numbered in lexicographic order of (z, y, x).  Tensor-ordered dofmap in
    each cell's own frame."""
    p = int(degree)
    x = np.ascontiguousarray(mesh.x, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    nd = (p + 1) ** 3
    dm = np.zeros((cells.shape[0], nd), dtype=np.int32)
    cap = cells.shape[0] * nd
    coords = np.zeros((max(cap, 1), 3))
    nd_out = c_int64(0)
    check(lib().wf_fs_build(p, x.shape[0], _dp(x), cells.shape[0], _ip(cells), ctypes.byref(nd_out), _ip(dm), _dp(coords), cap))
    V = FunctionSpace(mesh, p, dm, IndexMap(int(nd_out.value)), None, structured=False)
    V.dof_coordinates = coords[: nd_out.value].copy()
    return V


def locate_facets(mesh, tags: MeshTags, value: int):
    """The (cell, axis, side) of every facet carrying `value`: the facet's vertex set is
    matched against the faces of the cells (an exterior facet belongs to one cell)."""
    fv = np.ascontiguousarray(tags.facet_vertices[tags.values == value], dtype=np.int32)
    cells = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    nf = fv.shape[0]
    c, a, s_ = (np.zeros(nf, dtype=np.int32) for _ in range(3))
    try:
        check(lib().wf_fs_locate_facets(cells.shape[0], _ip(cells), nf, _ip(fv), _ip(c), _ip(a), _ip(s_)))
    except Exception as e:
        raise ValueError(str(e)) from e
    return list(zip(c.tolist(), a.tolist(), s_.tolist()))


def facet_lumped_mass(V: FunctionSpace, facets):
    """Collocated facet masses m[i] = sum_facets w_q |dx/ds x dx/dt| (diagonal GLL form of
    inner(g, v) * ds(tag), demo/cpu_planar3d/forms.ufl:19-24) for a list of (cell, axis, side).
    Returns (dof indices int32 ascending, masses)."""
    mesh, p = V.mesh, V.degree
    x = np.ascontiguousarray(mesh.x, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    dm = np.ascontiguousarray(V.dofmap, dtype=np.int32)
    f = np.asarray(list(facets), dtype=np.int32).reshape(-1, 3)
    fc, fa, fs = (np.ascontiguousarray(f[:, k]) for k in range(3))
    cap = max(1, f.shape[0] * (p + 1) ** 2)
    idx, m = np.zeros(cap, dtype=np.int32), np.zeros(cap)
    nout = c_int64(0)
    check(lib().wf_fs_facet_mass(p, x.shape[0], _dp(x), cells.shape[0], _ip(cells), _ip(dm), f.shape[0], _ip(fc), _ip(fa),
                                 _ip(fs), ctypes.byref(nout), _ip(idx), _dp(m)))
    return idx[: nout.value].copy(), m[: nout.value].copy()


def boundary_sets(V: FunctionSpace, tags: MeshTags, values=(1, 2)):
    """The `boundary=` argument of LinearGLLOpt from the file's facet tags
    (tag 1 = Gamma_1 source, tag 2 = Gamma_2 absorbing; common/LinearGLL.hpp:113-115)."""
    return tuple(facet_lumped_mass(V, locate_facets(V.mesh, tags, v)) for v in values)


def cfl_time_step(mesh, degree: int, c0: float, freq: float, CFL: float = 0.5):
    """demo/cpu_planar3d/main.cpp:48-66 (mesh::h = largest vertex distance of a cell)."""
    x = np.ascontiguousarray(mesh.x, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.geom_dofmap, dtype=np.int32)
    h = c_double(0.0)
    check(lib().wf_fs_min_cell_diameter(x.shape[0], _dp(x), cells.shape[0], _ip(cells), ctypes.byref(h)))
    dt = CFL * h.value / (c0 * degree ** 2)
    period = 1.0 / freq
    spp = int(period / dt + 1)
    return period / spp, spp


# ---------------------------------------------------------------------------
# synthetic meshes that are NOT a consistently oriented box (test / benchmark inputs: the
# reference's own mesh.xdmf is not in its repository)
# ---------------------------------------------------------------------------
_AXIS_PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def orientation_vertex_map(code: int) -> np.ndarray:
    """Vertex relabelling of one of the 48 orientations of the reference cube (code = 8 * perm +
    flips, the convention of csrc/generic_plan.cpp): new local vertex v' (bits along the new axes)
    is old vertex map[v']: new axis m runs along old axis perm[m], reversed if bit m of flips."""
    perm, flips = _AXIS_PERMS[code >> 3], code & 7
    out = np.zeros(8, dtype=np.int64)
    for v in range(8):
        old = 0
        for m in range(3):
            bit = ((v >> m) & 1) ^ ((flips >> m) & 1)
            old |= bit << perm[m]
        out[v] = old
    return out


def reorient_cells(mesh, cells, codes) -> BoxMesh:
    """The same mesh with the listed cells' local frames rotated / reflected (their vertices
    relabelled): what a multi-block or gmsh hexahedral mesh looks like.  codes: one orientation
    code (0..47) per listed cell, or one for all."""
    gd = np.array(mesh.geom_dofmap, dtype=np.int32, copy=True)
    cells = np.asarray(cells, dtype=np.int64)
    codes = np.broadcast_to(np.asarray(codes, dtype=np.int64), cells.shape)
    for code in np.unique(codes):
        sel = cells[codes == code]
        gd[sel] = gd[sel][:, orientation_vertex_map(int(code))]
    return BoxMesh(None, np.array(mesh.x, copy=True), gd, mesh.lo, mesh.hi)


def create_ogrid(m: int, nz: int, height: float = 1.0, perturb: float = 0.0, seed: int = 7) -> BoxMesh:
    """Three blocks of m x m x nz hexahedra around a common edge (a hexagonal prism cut into three
    quadrilateral prisms): the simplest mesh with an IRREGULAR edge (three cells around it), as in
    every O-grid.  No global lattice exists; each block is one."""
    ang = np.deg2rad(60.0 * np.arange(6))
    H = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    C = np.zeros(2)
    quads = [(C, H[0], H[1], H[2]), (C, H[2], H[3], H[4]), (C, H[4], H[5], H[0])]   # corners (0,0), (1,0), (1,1), (0,1)
    pts, cells = [], []
    t = np.linspace(0.0, 1.0, m + 1)
    zs = np.linspace(0.0, height, nz + 1)
    for q0, q1, q2, q3 in quads:
        base = len(pts)
        for k in range(nz + 1):
            for b in range(m + 1):
                for a in range(m + 1):
                    s_, r = t[a], t[b]
                    xy = (1 - s_) * (1 - r) * q0 + s_ * (1 - r) * q1 + s_ * r * q2 + (1 - s_) * r * q3
                    pts.append((xy[0], xy[1], zs[k]))
        vid = lambda a, b, k: base + a + (m + 1) * (b + (m + 1) * k)   # noqa: E731
        for k in range(nz):
            for b in range(m):
                for a in range(m):
                    cells.append([vid(a + (v & 1), b + ((v >> 1) & 1), k + ((v >> 2) & 1)) for v in range(8)])
    pts = np.asarray(pts)
    # merge the vertices the blocks share (generator only: exact copies of the same expressions up to rounding)
    key = np.round(pts / 1e-9).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    x = pts[first].copy()
    gd = inv.reshape(-1)[np.asarray(cells, dtype=np.int64)].astype(np.int32)
    if perturb > 0.0:
        rng = np.random.default_rng(seed)
        r = np.hypot(x[:, 0], x[:, 1])
        inner = (r > 1e-9) & (r < 0.8) & (x[:, 2] > 1e-9) & (x[:, 2] < height - 1e-9)
        x[inner] += rng.uniform(-1, 1, size=(int(inner.sum()), 3)) * (perturb / m) * np.array([1.0, 1.0, height * m / nz])
    return BoxMesh(None, x, gd)
