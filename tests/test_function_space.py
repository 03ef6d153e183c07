"""The function space of an arbitrary conforming hexahedral mesh (wf_fs_build, csrc/function_space.cpp;
fem::create_functionspace in demo/cpu_planar3d/main.cpp:36-45): dofs are identified topologically, so
cells that see a shared face, edge or vertex in different local frames agree EXACTLY -- comparing
coordinates (computed through differently ordered trilinear sums, ~1 ulp apart) must not decide
identity.  Host-only; also the lattice plan's orientation normalisation on the CPU side is covered
through the GPU tests (tests/test_gpu_unstructured.py)."""
import numpy as np
import pytest


def conformity_defect(mesh, V, oracle):
    """max over dofs of the spread of the physical coordinates the cells assign to it, and the smallest
    distance between two different dofs (must be far apart)."""
    p = V.degree
    X, _ = oracle.quadrature_weights_hex(p)
    phi, _ = oracle.cmap_tabulate(X)
    xd = np.einsum("qv,cvd->cqd", phi, mesh.x[mesh.geom_dofmap]).reshape(-1, 3)
    flat = V.dofmap.reshape(-1)
    lo = np.full((V.ndofs, 3), np.inf)
    hi = np.full((V.ndofs, 3), -np.inf)
    np.minimum.at(lo, flat, xd)
    np.maximum.at(hi, flat, xd)
    assert np.isfinite(lo).all(), "a dof number is not used by any cell"
    return float((hi - lo).max()), lo


@pytest.mark.parametrize("p,n", [(1, (6, 5, 4)), (2, (5, 4, 3)), (3, (4, 3, 3)), (4, (20, 20, 20)), (5, (3, 3, 2)), (7, (2, 2, 2))])
def test_randomly_reoriented_box_is_conforming(oracle, p, n):
    """Every cell of a perturbed box in a random one of the 48 orientations of the reference cube
    (the coordinate-rounding dofmap duplicated 8 shared dofs of a 40^3 P4 mesh)."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    box = w.create_box(n, perturb=0.2)
    rng = np.random.default_rng(11)
    mesh = mesh_io.reorient_cells(box, np.arange(box.ncells), rng.integers(0, 48, box.ncells))
    V = mesh_io.create_functionspace(mesh, p)
    assert V.ndofs == (p * n[0] + 1) * (p * n[1] + 1) * (p * n[2] + 1)
    assert V.dofmap.min() == 0 and V.dofmap.max() == V.ndofs - 1
    spread, X = conformity_defect(mesh, V, oracle)
    assert spread <= 1e-13
    assert np.abs(V.dof_coordinates - X).max() <= 1e-13
    # dofs are numbered in lexicographic (z, y, x) order of their coordinates (quantised to 1e-9 of the
    # shortest cell edge, which only decides the ORDER)
    xc = mesh.x[mesh.geom_dofmap]
    emin = min(np.linalg.norm(xc[:, 1 << d] - xc[:, 0], axis=1).min() for d in range(3))
    key = np.round(V.dof_coordinates / (1e-9 * emin)).astype(np.int64)
    order = np.lexsort((key[:, 0], key[:, 1], key[:, 2]))
    assert np.array_equal(key[order], key)
    # each cell's dofs are distinct
    assert all(np.unique(row).size == row.size for row in V.dofmap[:: max(1, mesh.ncells // 50)])


def test_consistent_box_gets_the_lattice_numbering():
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    for p in (1, 2, 4, 6):
        mesh = w.create_box((4, 3, 2))
        assert np.array_equal(mesh_io.create_functionspace(mesh, p).dofmap, w.create_functionspace(mesh, p).dofmap)


@pytest.mark.parametrize("p", [1, 2, 4])
def test_ogrid_counts(oracle, p):
    """Three blocks around an irregular edge: 3 (P m)^2 + 3 P m + 1 dofs per plane."""
    from wave_fenics_amd import mesh_io
    m, nz = 3, 2
    mesh = mesh_io.create_ogrid(m, nz, perturb=0.1)
    assert mesh.ncells == 3 * m * m * nz
    V = mesh_io.create_functionspace(mesh, p)
    assert V.ndofs == (3 * (p * m) ** 2 + 3 * p * m + 1) * (p * nz + 1)
    spread, _ = conformity_defect(mesh, V, oracle)
    assert spread <= 1e-13


def test_facets_masses_and_cfl_match_numpy():
    """wf_fs_locate_facets / wf_fs_facet_mass / wf_fs_min_cell_diameter against plain numpy on a
    perturbed box with every exterior facet tagged: masses sum to the face areas."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    n, p = (3, 2, 2), 3
    mesh = w.create_box(n, hi=(1.0, 0.5, 2.0))
    V = mesh_io.create_functionspace(mesh, p)
    fv, val = [], []
    nx, ny, nz = n
    for c in range(mesh.ncells):
        cx, cy, cz = c % nx, (c // nx) % ny, c // (nx * ny)
        for axis, (cc, nn) in enumerate(((cx, nx), (cy, ny), (cz, nz))):
            for side in (0, 1):
                if cc == (nn - 1 if side else 0):
                    lv = [v for v in range(8) if ((v >> axis) & 1) == side]
                    fv.append(mesh.geom_dofmap[c, lv][::-1])          # any vertex order
                    val.append(1 if (axis == 0 and side == 0) else 2)
    tags = mesh_io.MeshTags(np.array(fv, dtype=np.int32), np.array(val, dtype=np.int32))
    (i1, m1), (i2, m2) = mesh_io.boundary_sets(V, tags)
    assert abs(m1.sum() - 0.5 * 2.0) <= 1e-13
    assert abs(m2.sum() - (0.5 * 2.0 + 2 * 1.0 * 2.0 + 2 * 1.0 * 0.5)) <= 1e-12
    assert np.all(np.diff(i1) > 0) and np.all(np.diff(i2) > 0)
    dt, spp = mesh_io.cfl_time_step(mesh, p, 1500.0, 0.5e6)
    h = np.sqrt((1 / 3) ** 2 + 0.25 ** 2 + 1.0)
    period = 1 / 0.5e6
    assert spp == int(period / (0.5 * h / (1500.0 * p * p)) + 1) and abs(dt - period / spp) < 1e-20
    bad = mesh_io.MeshTags(np.array([[0, 1, 2, 99]], dtype=np.int32), np.array([1], dtype=np.int32))
    with pytest.raises(ValueError):
        mesh_io.locate_facets(mesh, bad, 1)


@pytest.mark.parametrize("p", [1, 2, 4])
def test_lattice_numbering_is_a_permutation_with_contiguous_rows(p):
    """wf_lattice_numbering (the setup-time renumbering option): from a uniformly random dof numbering of a box
    it returns a permutation under which every x-row of a cell column is contiguous, consecutive rows of a
    plane follow each other, and a mesh that does not tile still gets a valid (first-touch) permutation."""
    import wave_fenics_amd as w
    mesh = w.create_box((5, 4, 9))
    V = w.create_functionspace(mesh, p)
    rng = np.random.default_rng(p)
    scr = rng.permutation(V.ndofs).astype(np.int32)
    Vs = w.renumber(V, scr)
    new = w.lattice_numbering(Vs)
    assert sorted(new.tolist()) == list(range(V.ndofs))
    Vn = w.renumber(Vs, new)
    n = p + 1
    dm = Vn.dofmap.reshape(-1, n, n, n)            # [cell][k][j][i]
    d = np.diff(dm, axis=3)
    assert (d == 1).mean() > 0.9                    # rows contiguous (a row may cross into a face plane numbered by the neighbour column)
    # a cell's dofs span far less memory than under the scrambled numbering
    span_new = (dm.reshape(len(dm), -1).max(1) - dm.reshape(len(dm), -1).min(1)).mean()
    dms = Vs.dofmap.reshape(len(dm), -1)
    span_old = (dms.max(1) - dms.min(1)).mean()
    assert span_new < 0.25 * span_old
    # non-tiling input (two cells glued with inconsistent corner dofs): still a permutation
    bad = np.arange(2 * n ** 3, dtype=np.int32).reshape(2, -1)
    bad[1, 0] = bad[0, 0]
    Vb = w.FunctionSpace(mesh, p, bad, w.IndexMap(2 * n ** 3), (0, 0, 0), structured=False)
    nb = w.lattice_numbering(Vb)
    assert sorted(nb.tolist()) == list(range(2 * n ** 3))


def test_cxx_mesh_space_and_renumbering(tmp_path):
    """include/wavehip_mesh.hpp on the host only: function space of a scrambled box (wf_fs_build) and
    wavehip::renumber_lattice (wf_lattice_numbering) through the C++ wrappers."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import wave_fenics_amd as w
    w.lib()
    exe = str(tmp_path / "renumber_smoke")
    libdir = os.path.join(root, "wave_fenics_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cxx", "renumber_smoke.cpp"), "-o", exe, "-L", libdir, "-lwavehip",
                           f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert out.stdout.split() == [str(13 * 10 * 16), "ok"]
