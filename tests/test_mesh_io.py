"""Mesh / facet-tag input: the XDMF + HDF5 files of
demo/cpu_planar3d/main.cpp:39-45 through wf_mesh_* (csrc/mesh_io.cpp), the function space
and boundary sets derived from them, and the RK4 loop driven from a mesh FILE.
The reference's mesh.xdmf is not in its repository: pinned by round trips, by the
independent symbolic golden (facet masses) and by parity with the box meshes."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shuffled_box(oracle, n, p, hi, perturb, seed=5):
    """An oracle box mesh and the same mesh with cells and vertices renumbered at random and
    every exterior facet tagged as the oracle tags it (1: x = lo, 2: the rest)."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    om = oracle.create_box(n, p, hi=hi, perturb=perturb)
    rng = np.random.default_rng(seed)
    vperm = rng.permutation(om.x.shape[0])                 # new vertex id of old vertex v
    cperm = rng.permutation(om.ncells)                     # new cell order
    x2 = np.empty_like(om.x)
    x2[vperm] = om.x
    cells2 = vperm[om.geom_dofmap][cperm].astype(np.int32)
    mesh = w.BoxMesh(None, x2, np.ascontiguousarray(cells2))
    fv, val = [], []
    for cells, lf, tag in oracle.box_facets(om):
        axis, side = lf // 2, lf % 2
        lv = [v for v in range(8) if ((v >> axis) & 1) == side]
        for c in cells:
            fv.append(vperm[om.geom_dofmap[c, lv]])
            val.append(tag)
    order = rng.permutation(len(fv))
    tags = mesh_io.MeshTags(np.array(fv, dtype=np.int32)[order], np.array(val, dtype=np.int32)[order])
    return om, mesh, tags


def test_xdmf_round_trip_and_errors(oracle, tmp_path):
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    om, mesh, tags = shuffled_box(oracle, (3, 2, 2), 2, (1.0, 1.0, 1.0), 0.2)
    path = str(tmp_path / "mesh.xdmf")
    mesh_io.write_mesh(path, "planar3d", mesh, "planar3d_boundaries", tags)
    assert os.path.exists(str(tmp_path / "mesh.h5"))
    m2, t2 = mesh_io.read_mesh(path, "planar3d", "planar3d_boundaries")
    assert np.array_equal(m2.x, mesh.x) and np.array_equal(m2.geom_dofmap, mesh.geom_dofmap)
    assert np.array_equal(t2.facet_vertices, tags.facet_vertices) and np.array_equal(t2.values, tags.values)
    m3 = mesh_io.read_mesh(path, "planar3d")
    assert np.array_equal(m3.geom_dofmap, mesh.geom_dofmap)
    with pytest.raises(w.WavehipError):
        mesh_io.read_mesh(path, "no_such_grid")
    with pytest.raises(w.WavehipError):
        mesh_io.read_mesh(str(tmp_path / "missing.xdmf"), "planar3d")
    with pytest.raises(w.WavehipError):
        mesh_io.read_mesh(path, "planar3d", "no_such_tags")


@pytest.mark.parametrize("p", [1, 2, 4])
def test_function_space_from_mesh(oracle, p):
    """The coordinate-derived dofmap of a renumbered box equals the box dofmap up to a
    renumbering of the dofs, cell by cell and position by position."""
    from wave_fenics_amd import mesh_io
    om, mesh, tags = shuffled_box(oracle, (3, 2, 2), p, (1.0, 0.7, 1.3), 0.2)
    V = mesh_io.create_functionspace(mesh, p)
    assert V.ndofs == om.ndofs
    X = oracle.dof_coordinates(om)
    # cell c of `mesh` is cell cperm[c] of the oracle mesh: recover the correspondence through the vertices
    key = {tuple(np.round(v, 12)): i for i, v in enumerate(om.x)}
    vmap = np.array([key[tuple(np.round(v, 12))] for v in mesh.x])
    ocell = {tuple(om.geom_dofmap[c]): c for c in range(om.ncells)}
    for c in range(mesh.ncells):
        oc = ocell[tuple(vmap[mesh.geom_dofmap[c]])]
        assert np.abs(V.dof_coordinates[V.dofmap[c]] - X[om.dofmap[oc]]).max() <= 1e-12
    # a bijection old dof -> new dof
    pairs = set()
    for c in range(mesh.ncells):
        oc = ocell[tuple(vmap[mesh.geom_dofmap[c]])]
        pairs.update(zip(om.dofmap[oc].tolist(), V.dofmap[c].tolist()))
    assert len(pairs) == om.ndofs


def test_boundary_sets_from_tags_vs_independent_golden():
    """Facet tags given as vertex quadruples -> (cell, face) -> collocated facet masses, on the
    meshes of the symbolic golden (non-rectangular boundary facets)."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    with open(os.path.join(ROOT, "tests", "golden", "independent.json")) as f:
        golden = json.load(f)
    dec = lambda e: np.array([float(v) for v in e["data"]]).reshape(e["shape"])
    for case in golden["mesh_cases"]:
        p, n = case["p"], tuple(case["n"])
        box = w.create_box(n)
        box.x = dec(case["verts"]).copy()
        fv, val = [], []
        nx, ny, nz = n
        for c in range(box.ncells):
            cx, cy, cz = c % nx, (c // nx) % ny, c // (nx * ny)
            for axis, (cc, nn) in enumerate(((cx, nx), (cy, ny), (cz, nz))):
                for side in (0, 1):
                    if cc == (0 if side == 0 else nn - 1):
                        fv.append(box.geom_dofmap[c, [v for v in range(8) if ((v >> axis) & 1) == side]])
                        val.append(1 if (axis == 0 and side == 0) else 2)
        tags = mesh_io.MeshTags(np.array(fv, dtype=np.int32), np.array(val, dtype=np.int32))
        V = w.create_functionspace(box, p)            # box numbering, so that the golden's dof order applies
        for tag, key in ((1, "mG1"), (2, "mG2")):
            idx, m = mesh_io.facet_lumped_mass(V, mesh_io.locate_facets(box, tags, tag))
            dense = np.zeros(V.ndofs)
            dense[idx] = m
            ref = dec(case[key])
            assert np.abs(dense - ref).max() <= 1e-14 * ref.max()


@pytest.mark.gpu
def test_rk4_from_mesh_file(oracle, tmp_path):
    """demo/cpu_planar3d's flow with the mesh coming from a FILE: write a perturbed box whose
    cells and vertices are renumbered at random (XDMF + HDF5, tagged facets), read it back,
    derive function space and boundary sets, run 20 RK4 steps on the MI355X through the
    generic operators, and compare with the oracle on the original box (dofs matched by
    coordinates)."""
    import torch
    from wave_fenics_amd import mesh_io
    from wave_fenics_amd.linear_gll import LinearGLLOpt
    p, n, hi = 3, (4, 3, 3), (0.01, 0.0075, 0.0075)
    om, mesh, tags = shuffled_box(oracle, n, p, hi, 0.15)
    path = str(tmp_path / "mesh.xdmf")
    mesh_io.write_mesh(path, "planar3d", mesh, "planar3d_boundaries", tags)
    m2, t2 = mesh_io.read_mesh(path, "planar3d", "planar3d_boundaries")
    V = mesh_io.create_functionspace(m2, p)
    sets = mesh_io.boundary_sets(V, t2)
    dt, spp = mesh_io.cfl_time_step(m2, p, 1500.0, 0.5e6, CFL=0.25)
    dto, sppo = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.25)
    assert dt == dto and spp == sppo
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    ref.init()
    ref.rk4(0.0, 20 * dt - 1e-13, dt)
    eqn = LinearGLLOpt(V, p, 1500.0, 0.5e6, 6e4, boundary=sets, device=torch.device("cuda", 0), structured=False)
    eqn.init()
    eqn.rk4_fused(0.0, 20 * dt - 1e-13, dt)
    # match dofs by coordinates
    X = oracle.dof_coordinates(om)
    e = np.linalg.norm(om.x[om.geom_dofmap[:, 1]] - om.x[om.geom_dofmap[:, 0]], axis=1).min()
    qa = np.round(X / (1e-9 * e)).astype(np.int64)
    qb = np.round(V.dof_coordinates / (1e-9 * e)).astype(np.int64)
    ia = np.lexsort(qa.T[::-1])
    ib = np.lexsort(qb.T[::-1])
    assert np.array_equal(qa[ia], qb[ib])
    u, v = eqn.u_n.cpu().numpy(), eqn.v_n.cpu().numpy()
    assert np.abs(u[ib] - ref.u_n[ia]).max() <= 1e-9 * np.abs(ref.u_n).max()
    assert np.abs(v[ib] - ref.v_n[ia]).max() <= 1e-9 * np.abs(ref.v_n).max()

@pytest.mark.gpu
def test_rk4_from_mesh_file_cxx(oracle, tmp_path):
    """The same flow through the C++ host code (include/wavehip_mesh.hpp + the Space constructor of
    wavehip::LinearGLLOpt, examples/planar3d --mesh FILE): mesh and facet tags read from the file in
    C++, function space / boundary sets / CFL step derived in C++, fused and reference-order RK4,
    with roctx range markers on -- against the oracle on the original box
    (demo/cpu_planar3d/main.cpp:39-66)."""
    import subprocess
    from wave_fenics_amd import mesh_io
    p, n, hi = 3, (4, 3, 3), (0.01, 0.0075, 0.0075)
    om, mesh, tags = shuffled_box(oracle, n, p, hi, 0.15)
    path = str(tmp_path / "mesh.xdmf")
    mesh_io.write_mesh(path, "planar3d", mesh, "planar3d_boundaries", tags)
    out = str(tmp_path / "bin")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples"), f"OUT={out}", f"{out}/planar3d"])
    dt, spp = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.25)
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    ref.init()
    ref.rk4(0.0, 20 * dt - 1e-13, dt)
    V = mesh_io.create_functionspace(mesh_io.read_mesh(path, "planar3d"), p)     # the numbering the C++ side derives too
    X = oracle.dof_coordinates(om)
    e = np.linalg.norm(om.x[om.geom_dofmap[:, 1]] - om.x[om.geom_dofmap[:, 0]], axis=1).min()
    qa = np.round(X / (1e-9 * e)).astype(np.int64)
    qb = np.round(V.dof_coordinates / (1e-9 * e)).astype(np.int64)
    ia, ib = np.lexsort(qa.T[::-1]), np.lexsort(qb.T[::-1])
    assert np.array_equal(qa[ia], qb[ib])
    for extra in (["--markers"], ["--reference-order"]):
        dump = str(tmp_path / "uv.bin")
        r = subprocess.run([os.path.join(out, "planar3d"), "--mesh", path, "--degree", str(p), "--cfl", "0.25", "--steps", "20",
                            "--dump", dump] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Steps taken: 20" in r.stdout and f"Degrees of freedom: {om.ndofs}" in r.stdout
        assert f"Number of step per period: {spp}" in r.stdout
        uv = np.fromfile(dump, dtype=np.float64)
        assert uv.size == 2 * om.ndofs
        u, v = uv[: om.ndofs], uv[om.ndofs:]
        assert np.abs(u[ib] - ref.u_n[ia]).max() <= 1e-9 * np.abs(ref.u_n).max()
        assert np.abs(v[ib] - ref.v_n[ia]).max() <= 1e-9 * np.abs(ref.v_n).max()


def test_tag_reader_checks_extents_and_ranges(tmp_path):
    """wf_mesh_read_tags must not trust the file -- a Values dataset with another element
    count than the tag topology, or a facet vertex id outside the mesh, is an error, never a buffer
    overrun; a grid whose geometry lives in another .h5 than its topology is refused.  (The files are
    doctored by pointing the XDMF text at datasets of a second file written by wf_mesh_write.)"""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    mesh = w.create_box((2, 1, 1))
    one = mesh_io.MeshTags(np.array([[0, 1, 3, 4]], dtype=np.int32), np.array([1], dtype=np.int32))
    three = mesh_io.MeshTags(np.array([[0, 1, 3, 4], [1, 2, 4, 5], [0, 3, 6, 9]], dtype=np.int32), np.array([1, 2, 2], dtype=np.int32))
    a, b = str(tmp_path / "a.xdmf"), str(tmp_path / "b.xdmf")
    mesh_io.write_mesh(a, "planar3d", mesh, "planar3d_boundaries", one)
    mesh_io.write_mesh(b, "planar3d", mesh, "planar3d_boundaries", three)
    txt = open(a).read()
    # (1) Values from the other file: 3 values for 1 tagged facet
    bad = str(tmp_path / "bad_values.xdmf")
    open(bad, "w").write(txt.replace("a.h5:/MeshTags/planar3d_boundaries/Values", "b.h5:/MeshTags/planar3d_boundaries/Values"))
    with pytest.raises(w.WavehipError, match="one value per tagged facet"):
        mesh_io.read_mesh(bad, "planar3d", "planar3d_boundaries")
    # (2) a facet vertex id beyond the mesh's vertices
    big = mesh_io.MeshTags(np.array([[0, 1, 3, 400]], dtype=np.int32), np.array([1], dtype=np.int32))
    c = str(tmp_path / "c.xdmf")
    mesh_io.write_mesh(c, "planar3d", mesh, "planar3d_boundaries", big)
    with pytest.raises(w.WavehipError, match="facet vertex index out of range"):
        mesh_io.read_mesh(c, "planar3d", "planar3d_boundaries")
    # (3) geometry in another file than the topology
    split = str(tmp_path / "split.xdmf")
    open(split, "w").write(txt.replace("a.h5:/Mesh/planar3d/geometry", "b.h5:/Mesh/planar3d/geometry"))
    with pytest.raises(w.WavehipError, match="different HDF5 files"):
        mesh_io.read_mesh(split, "planar3d")
    # the untouched files still read, and the mesh can be re-read after its tags (the handle reopens the mesh file)
    m2, t2 = mesh_io.read_mesh(b, "planar3d", "planar3d_boundaries")
    assert np.array_equal(t2.values, three.values) and np.array_equal(m2.geom_dofmap, mesh.geom_dofmap)
