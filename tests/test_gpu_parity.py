"""GPU parity tests: libwavehip (through the C ABI, via wave_fenics_amd) against
the CPU oracle on the same seeded inputs.  Run on the GPU box: pytest -m gpu.

Tolerance (SURVEY.md 8c / BASELINE.json): fp64; single operator apply
  max|y_gpu - y_ref| <= 1e-12 * max|y_ref|
(dense reference summation order vs sum-factorised order + atomic ordering);
integer/index work (gather) is bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import wave_fenics_amd as w
    w.lib()
    return torch.device("cuda", 0)


def dev(a, gpu, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    return t if dtype is None else t.to(dtype)


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make(oracle, n, p, perturb=0.2):
    import wave_fenics_amd as w
    om = oracle.create_box(n, p, perturb=perturb)
    mesh = w.create_box(n, perturb=perturb)
    V = w.create_functionspace(mesh, p)
    assert np.array_equal(V.dofmap, om.dofmap) and np.array_equal(mesh.x, om.x)
    return om, mesh, V


@pytest.mark.parametrize("p,n", [(1, (5, 4, 3)), (2, (4, 3, 3)), (3, (3, 3, 2)), (4, (3, 2, 2)), (4, (6, 5, 1)),
                                 (5, (2, 2, 2)), (6, (2, 2, 1)), (7, (2, 1, 1))])
def test_stiffness_generic_vs_oracle(gpu, oracle, p, n):
    import wave_fenics_amd as w
    om, mesh, V = make(oracle, n, p)
    K = oracle.StiffnessOperator(om, p)
    rng = np.random.default_rng(1234)
    x = rng.uniform(-1, 1, om.ndofs)
    y0 = rng.uniform(-1, 1, om.ndofs) * 1e6          # accumulate semantics: y += K x
    yref = y0.copy()
    K(x, yref)
    # (a) geometry handed over in the reference layout (isolates the kernel)
    op = w.StiffnessOperator(V, p, {"c0": 1500.0}, G=K.G, structured=False)
    y = dev(y0, gpu)
    op(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= TOL
    # (b) geometry computed on the device from the mesh
    op2 = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=False)
    y2 = dev(y0, gpu)
    op2(dev(x, gpu), y2)
    assert relerr(y2.cpu().numpy(), yref) <= 1e-12
    assert op.num_cells() == om.ncells and op.num_dofs() == (p + 1) ** 3 and op.num_quads() == (p + 1) ** 3


def test_stiffness_generic_permuted_dofmap(gpu, oracle):
    """Arbitrary (non-lexicographic) global numbering and a non-identity
    element permutation, as a DOLFINx dofmap would present them
    (common/permute.hpp:10-27)."""
    import wave_fenics_amd as w
    p, n = 3, (3, 2, 2)
    om, mesh, V = make(oracle, n, p)
    rng = np.random.default_rng(7)
    gperm = rng.permutation(om.ndofs).astype(np.int32)        # renumber global dofs
    eperm = rng.permutation((p + 1) ** 3).astype(np.int32)    # tensor -> element order
    inv = np.empty_like(eperm)
    inv[eperm] = np.arange(eperm.size, dtype=np.int32)
    dm_elem = gperm[om.dofmap][:, inv]                         # element-ordered dofmap
    assert np.array_equal(dm_elem[:, eperm], gperm[om.dofmap])
    K = oracle.StiffnessOperator(om, p)
    x = rng.uniform(-1, 1, om.ndofs)
    yref = np.zeros(om.ndofs)
    K(x, yref)
    Vp = w.FunctionSpace(mesh, p, dm_elem.astype(np.int32), w.IndexMap(om.ndofs), V.lattice, structured=False)
    op = w.StiffnessOperator(Vp, p, G=K.G, perm=eperm)
    xp = np.empty_like(x)
    xp[gperm] = x
    y = dev(np.zeros(om.ndofs), gpu)
    op(dev(xp, gpu), y)
    assert relerr(y.cpu().numpy()[gperm], yref) <= TOL


@pytest.mark.parametrize("p,n,block", [(4, (5, 4, 3), "5,2,1"), (4, (5, 4, 3), "2,2,2"), (4, (7, 3, 2), "3,3,1"),
                                       (4, (4, 4, 4), "1,1,1"), (2, (7, 5, 4), "3,3,3"), (3, (5, 4, 3), "4,2,2"),
                                       (1, (9, 5, 5), "4,4,4"), (5, (3, 2, 2), "7,1,1"), (6, (3, 2, 2), "5,1,1"),
                                       (7, (2, 2, 1), "2,2,1")])
def test_stiffness_box_block_vs_oracle(gpu, oracle, p, n, block):
    """Structured (implicit dofmap) single-pass block kernel incl. partial blocks
    at the mesh end (selected through wf_tuning)."""
    import wave_fenics_amd as w
    tuning = {"kernel": "box_block", "block": tuple(int(v) for v in block.split(","))}
    om, mesh, V = make(oracle, n, p)
    K = oracle.StiffnessOperator(om, p)
    rng = np.random.default_rng(99)
    x = rng.uniform(-1, 1, om.ndofs)
    y0 = rng.uniform(-1, 1, om.ndofs) * 1e6
    yref = y0.copy()
    K(x, yref)
    op = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=True, tuning=tuning)
    assert op.info.structured == 1 and op.kernel == "box_block"
    y = dev(y0, gpu)
    op(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= 1e-12


@pytest.mark.parametrize("p,n,variant,lz", [
    (4, (7, 5, 6), 0, 2), (4, (7, 5, 6), 1, 4), (4, (3, 3, 5), 2, 1), (4, (6, 6, 7), 0, 100), (4, (4, 2, 3), 1, 3),
    (1, (9, 9, 5), 0, 2), (1, (5, 4, 3), 1, 1), (1, (17, 5, 4), 2, 3),
    (2, (7, 6, 5), 0, 2), (2, (4, 4, 4), 1, 3), (2, (8, 5, 3), 2, 2),
    (3, (5, 5, 4), 0, 3), (3, (4, 4, 3), 1, 2), (3, (5, 3, 3), 2, 1),
    (4, (7, 5, 6), 3, 2), (4, (11, 2, 5), 3, 4),
    (5, (4, 3, 3), 0, 2), (5, (3, 3, 2), 1, 1), (5, (8, 2, 2), 2, 2),
    (6, (3, 3, 3), 0, 2), (6, (6, 2, 2), 1, 1), (6, (4, 2, 2), 2, 3),
    (7, (3, 3, 2), 0, 1), (7, (5, 2, 2), 1, 2), (7, (3, 2, 3), 2, 2)])
def test_stiffness_march_vs_oracle(gpu, oracle, p, n, variant, lz):
    """The production box kernels (marching columns): every compiled column cross-section of the
    one-thread-per-column kernel (P <= 4, variants 0..2) and of the k-split kernel (P >= 5, and
    P4 as variant 3), partial columns at the mesh end, z segments of 1..all layers, accumulate
    semantics."""
    import wave_fenics_amd as w
    KS = {4: [(5, 1), (5, 2)], 5: [(3, 1), (7, 1), (2, 1)], 6: [(1, 1), (2, 1), (5, 1)], 7: [(2, 1), (2, 2), (1, 1)]}
    tune = {"variant": variant}
    if p >= 5:
        tune = {"block": KS[p][variant] + (1,)}
    elif variant == 3:
        tune = {"variant": 3, "block": KS[p][lz % 2] + (1,)}
    om, mesh, V = make(oracle, n, p)
    K = oracle.StiffnessOperator(om, p)
    rng = np.random.default_rng(2024)
    x = rng.uniform(-1, 1, om.ndofs)
    y0 = rng.uniform(-1, 1, om.ndofs) * 1e6
    yref = y0.copy()
    K(x, yref)
    op = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=True, tuning=dict(tune, lz=lz))
    assert op.kernel == "march_box" and op.info.plan_lz == lz
    y = dev(y0, gpu)
    op(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= 1e-12
    # default segmentation as well
    op = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=True, tuning=tune)
    y = dev(y0, gpu)
    op(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= 1e-12


@pytest.mark.parametrize("p,n", [(4, (11, 5, 6)), (6, (6, 2, 5))])
@pytest.mark.parametrize("structured", [True, False])
@pytest.mark.parametrize("ghost", [(1, 0, 0), (0, 1, 1), (1, 1, 1), (0, 0, 0)])
def test_interior_interface_split(gpu, oracle, ghost, structured, p, n):
    """apply(INTERIOR) + apply(INTERFACE) == apply, and the interior part reads no ghost dof of x
    (poisoned with NaN) -- for the box kernels (wf_op_set_ghost_faces and wf_op_set_ghost_dofs) and
    for the arbitrary-dofmap marching kernels (wf_op_set_ghost_dofs: an item is interface iff its
    dof tile contains a ghost position).  VectorUpdater.hpp:106-143,157-199."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd._lib import WF_PART_INTERFACE, WF_PART_INTERIOR, WF_PART_INTERIOR_A, WF_PART_INTERIOR_B
    om, mesh, V = make(oracle, n, p)
    NX, NY, NZ = V.lattice
    lat = np.arange(om.ndofs).reshape(NZ, NY, NX)
    gpos = np.concatenate([lat[:, :, 0].ravel() if ghost[0] else [], lat[:, 0, :].ravel() if ghost[1] else [],
                           lat[0, :, :].ravel() if ghost[2] else []]).astype(np.int32)
    rng = np.random.default_rng(8)
    x = dev(rng.uniform(-1, 1, om.ndofs), gpu)
    modes = ("faces", "dofs") if structured else ("dofs",)
    for mode in modes:
        op = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=structured, tuning={"lz": 2, "kernel": "march"})
        assert op.kernel == ("march_box" if structured else "march_idx")
        if mode == "faces":
            assert op.set_ghost_faces(*[bool(g) for g in ghost])
        else:
            assert op.set_ghost_dofs(gpos)
        assert op.info.items_interior + op.info.items_interface > 0
        if any(ghost):
            assert op.info.items_interface > 0 and op.info.items_interior > 0
        yall = torch.zeros_like(x)
        op(x, yall)
        ya = torch.zeros_like(x)
        xp = x.clone()
        xp[torch.from_numpy(gpos.astype(np.int64)).to(gpu)] = float("nan")
        op.apply_part(xp, ya, WF_PART_INTERIOR)
        assert bool(torch.isfinite(ya).all()), "interior part read a ghost dof"
        op.apply_part(x, ya, WF_PART_INTERFACE)
        assert relerr(ya.cpu().numpy(), yall.cpu().numpy()) <= 1e-13
        # the two interior halves (forward halo under A, reverse halo under B) add up to INTERIOR
        yb = torch.zeros_like(x)
        for part in (WF_PART_INTERIOR_A, WF_PART_INTERFACE, WF_PART_INTERIOR_B):
            op.apply_part(x, yb, part)
        assert relerr(yb.cpu().numpy(), yall.cpu().numpy()) <= 1e-13
        if ghost == (0, 0, 0):
            assert op.info.items_interface == 0
    # an operator on a batch kernel has no work items to sort: the caller uses the unsplit sequence
    opg = w.StiffnessOperator(V, p, structured=False, tuning={"kernel": "batch"})
    assert opg.kernel == "batch_unique" and opg.set_ghost_dofs(gpos) is False
    assert opg.set_ghost_faces(True, False, False) is False


def test_geometry_vs_oracle(gpu, oracle):
    import wave_fenics_amd as w
    for p, n in [(2, (3, 3, 2)), (4, (2, 2, 2))]:
        om, mesh, V = make(oracle, n, p)
        Gr, dr = oracle.precompute_geometric_data(om, p)
        G, d = w.precompute_geometric_data(mesh, p)
        assert relerr(d, dr) <= 1e-14
        assert np.abs(G - Gr).max() <= 1e-14 * np.abs(Gr).max()
        # affine mesh: exact zeros off the diagonal after the clamp
        om, mesh, V = make(oracle, n, p, perturb=0.0)
        G, d = w.precompute_geometric_data(mesh, p)
        assert np.all(G[:, :, 0, 1] == 0.0) and np.all(G[:, :, 1, 2] == 0.0)


@pytest.mark.parametrize("p,n", [(2, (4, 3, 3)), (4, (3, 2, 2)), (6, (2, 2, 1))])
def test_lumped_mass_vs_oracle(gpu, oracle, p, n):
    import wave_fenics_amd as w
    om, mesh, V = make(oracle, n, p)
    M = oracle.MassOperatorCPU(om, p)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, om.ndofs)
    y0 = rng.uniform(-1, 1, om.ndofs)
    yref = y0.copy()
    M(x, yref)
    for structured in (False, True):
        op = w.MassOperatorLumped(V, p, structured=structured)
        y = dev(y0, gpu)
        op(dev(x, gpu), y)
        assert relerr(y.cpu().numpy(), yref) <= 1e-13
    # handed-over detJ (reference layout): pre-assembled diagonal (default) and the
    # reference's element-wise gather * detJ -> scatter-add order (spectral_mass.hpp:84-89)
    from wave_fenics_amd import _lib
    for flags in (0, _lib.WF_FLAG_MASS_ELEMENTWISE):
        op = w.MassOperatorLumped(V, p, detJ=M.detJ, structured=False, flags=flags)
        y = dev(y0, gpu)
        op.apply(dev(x, gpu), y)
        assert relerr(y.cpu().numpy(), yref) <= 1e-14
        assert op.alg_bytes() == (24.0 * om.ndofs if flags == 0 else om.ncells * (8.0 + 4.0) * (p + 1) ** 3 + 16.0 * om.ndofs)


def test_spectral_mass_x_ones(gpu, oracle):
    """The reference's own check procedure: x = 1, |y_gpu - y_cpu| <= 1e-8 abs
    (demo/gpu_operator_monolithic/main.cpp:102-118) -- here to 1e-13 relative."""
    import wave_fenics_amd as w
    p, n = 6, (2, 2, 2)
    om, mesh, V = make(oracle, n, p)
    M = oracle.MassOperatorCPU(om, p)
    yref = np.zeros(om.ndofs)
    M(np.ones(om.ndofs), yref)
    for structured in (False, True):
        op = w.SpectralMassOperator(V, p, structured=structured)
        y = dev(np.zeros(om.ndofs), gpu)
        op.apply(dev(np.ones(om.ndofs), gpu), y)
        got = y.cpu().numpy()
        assert np.abs(got - yref).max() <= 1e-8
        assert relerr(got, yref) <= 1e-13
    with pytest.raises(w.WavehipError):
        w.SpectralMassOperator(V, 1)


@pytest.mark.parametrize("p,variant,quad,qd", [(1, "gll", "gll", 1), (2, "gll", "gll", 3), (3, "gll", "gll", 4),
                                                (2, "equispaced", "gauss_jacobi", 4),
                                                (3, "equispaced", "gauss_jacobi", 6),
                                                (4, "equispaced", "gauss_jacobi", 8)])
def test_dense_mass_vs_oracle(gpu, oracle, p, variant, quad, qd):
    """MassOperator (common/cuda/mass.hpp) incl. the non-collocated cuBLAS-demo
    configuration (demo/gpu_operator/main.cpp:66-68,96-99)."""
    import wave_fenics_amd as w
    n = (3, 2, 2)
    om, mesh, V = make(oracle, n, p)
    pts, wts, phi1, phi, X, W = oracle.tabulate_mass_tables(p, variant, quad, qd)
    detJ = oracle.compute_detJ_generic(om, X, W)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, om.ndofs)
    yref = np.zeros(om.ndofs)
    oracle.dense_mass_apply(om, phi, detJ, x, yref)
    op = w.MassOperator(V, p, phi1, detJ)
    y = dev(np.zeros(om.ndofs), gpu)
    op.apply(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= TOL
    assert op.flops() == 4.0 * om.ncells * phi.shape[0] * phi.shape[1]
    # a13 in the product: the reference's argument list MassOperator(V, element, quad_type, qd)
    # (mass.hpp:20-21) -- rule, 1-D table and det J * w at the rule's points all built by
    # libwavehip (wf_quadrature_1d, wf_tabulate_1d, device geometry), nothing taken from oracle/
    op2 = w.MassOperator(V, p, variant=variant, quad=quad, qdegree=qd)
    assert np.abs(op2.points1 - pts).max() <= 3e-16 and np.abs(op2.weights1 - wts).max() <= 1e-15
    y2 = dev(np.zeros(om.ndofs), gpu)
    op2.apply(dev(x, gpu), y2)
    assert relerr(y2.cpu().numpy(), yref) <= TOL
    assert op2.num_quads() == phi.shape[0]
    G, dj = w.compute_geometry_rule(mesh, pts, wts, want_G=True)       # precompute.hpp:49-176
    assert relerr(dj, detJ) <= 1e-14
    assert np.abs(G - np.swapaxes(G, 2, 3)).max() <= 1e-14 * np.abs(G).max()


@pytest.mark.parametrize("p,block,lz", [(2, None, 0), (2, None, 5), (3, None, 4), (4, None, 0), (4, (2, 2), 13), (4, None, 3),
                                        (5, None, 6), (6, None, 0), (6, (2, 1), 5), (7, None, 13)])
def test_dense_mass_long_columns(gpu, oracle, p, block, lz):
    """k_mass_march on columns of 13 layers: the index table streams through its LDS ring (4 P + 1 planes, so it
    wraps from the fourth layer on), the flush of a layer runs inside the next one, and the last segment is
    shorter than the others -- both compiled cross-sections of P4 / P6, several segment lengths."""
    import wave_fenics_amd as w
    n = (3, 2, 13)
    om, mesh, V = make(oracle, n, p)
    pts, wts, phi1, phi, X, W = oracle.tabulate_mass_tables(p, "equispaced", "gauss_jacobi", 2 * p)
    detJ = oracle.compute_detJ_generic(om, X, W)
    rng = np.random.default_rng(p + lz)
    x = rng.uniform(-1, 1, om.ndofs)
    yref = np.zeros(om.ndofs)
    oracle.dense_mass_apply(om, phi, detJ, x, yref)
    tun = {"lz": lz, "kernel": "march"}     # (a 3 x 2 column fills less than half of the low degrees' cross-sections)
    if block:
        tun["block"] = (*block, 1)
    op = w.MassOperator(V, p, phi1, detJ, tuning=tun)
    assert op.kernel == "march_idx" and (lz == 0 or op.info.plan_lz == lz)
    y = dev(np.zeros(om.ndofs), gpu)
    op.apply(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= TOL
    op.apply(dev(x, gpu), y)     # y += : a second apply doubles the result (the LDS tiles start from zero again)
    assert relerr(y.cpu().numpy(), 2 * yref) <= TOL


@pytest.mark.parametrize("p,n", [(1, (9, 9, 9)), (2, (9, 5, 7)), (3, (5, 5, 5)), (4, (5, 3, 9)), (6, (3, 3, 5))])
def test_marching_kernels_are_repeatable(gpu, p, n):
    """The cells of a layer are summed in an LDS tile (ds_add_f64) and the work items add to y with global atomics:
    the order of the additions differs from run to run, the data must not.  100 applies of the stiffness operator
    (box and arbitrary dofmap) and of the dense mass on a perturbed mesh agree to rounding (1e-14 of the largest
    entry); a race in the double-buffered LDS tiles or the index ring would show as an occasional O(1) deviation."""
    import torch
    import wave_fenics_amd as w
    mesh = w.create_box(n, perturb=0.2)
    V = w.create_functionspace(mesh, p)
    ops = [w.StiffnessOperator(V, p, structured=True),
           w.StiffnessOperator(V, p, structured=False, tuning={"kernel": "march"}),
           w.MassOperator(V, p, variant="equispaced", quad="gauss_jacobi", qdegree=2 * p, tuning={"kernel": "march"})]
    x = torch.rand(V.ndofs, dtype=torch.float64, device=gpu)
    for op in ops:
        ref, worst = None, 0.0
        for _ in range(100):
            y = torch.zeros_like(x)
            op.apply(x, y) if hasattr(op, "apply") else op(x, y)
            if ref is None:
                ref = y.clone()
            else:
                worst = max(worst, float((y - ref).abs().max() / ref.abs().max()))
        assert worst <= 1e-14, (p, op.kernel, worst)


@pytest.mark.parametrize("p", [2, 3, 4])
def test_x_slowest_tensor_order(gpu, oracle, p):
    """The DOLFINx-facing axis-order hazard: a caller whose tensor index is x-SLOWEST
    (Basix' tensor-product factorisation, l' = (i n + j) n + k) hands over its dofmap,
    an element permutation defined on that order, and G / detJ with points in that
    order.  With WF_FLAG_TENSOR_X_SLOWEST the result must equal the dense oracle on an
    anisotropic perturbed mesh (where pairing the wrong axes shows at once); without the
    flag the same inputs must NOT match (the hazard is real, the test can see it)."""
    import wave_fenics_amd as w
    from wave_fenics_amd import _lib
    n = (3, 2, 2)
    hi = (1.0, 0.37, 2.3)                                   # anisotropic: G_xx != G_yy != G_zz
    om = oracle.create_box(n, p, hi=hi, perturb=0.2)
    mesh = w.BoxMesh(om.n, om.x.copy(), om.geom_dofmap.copy(), (0.0, 0.0, 0.0), hi)
    nn = p + 1
    nd = nn ** 3
    k, j, i = np.meshgrid(np.arange(nn), np.arange(nn), np.arange(nn), indexing="ij")
    fast = (i + nn * (j + nn * k)).reshape(-1)              # engine index of (i, j, k)
    slow = ((i * nn + j) * nn + k).reshape(-1)              # Basix-style index of the same (i, j, k)
    to_slow = np.empty(nd, dtype=np.int64)
    to_slow[slow] = fast                                    # to_slow[l'] = l
    rng = np.random.default_rng(12)
    eperm = rng.permutation(nd).astype(np.int32)            # element (Basix dof) order <-> x-slowest tensor order
    # caller's arrays: dofmap in element order with dm_elem[c][eperm[l']] = dm_tensor_xslow[c][l']
    dm_xs = om.dofmap[:, to_slow]
    dm_elem = np.empty_like(dm_xs)
    dm_elem[:, eperm] = dm_xs
    K = oracle.StiffnessOperator(om, p)
    M = oracle.MassOperatorCPU(om, p)
    G_xs = np.ascontiguousarray(K.G[:, to_slow])            # point index in x-slowest order, 3x3 axes unchanged
    detJ_xs = np.ascontiguousarray(M.detJ[:, to_slow])
    x = rng.uniform(-1, 1, om.ndofs)
    yK, yM = np.zeros(om.ndofs), np.zeros(om.ndofs)
    K(x, yK)
    M(x, yM)
    V = w.FunctionSpace(mesh, p, np.ascontiguousarray(dm_elem), w.IndexMap(om.ndofs), om.lattice, structured=False)
    F = _lib.WF_FLAG_TENSOR_X_SLOWEST
    for kw in (dict(G=G_xs, perm=eperm), dict(perm=eperm)):            # handed-over G / device geometry
        y = dev(np.zeros(om.ndofs), gpu)
        w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=False, flags=F, **kw)(dev(x, gpu), y)
        assert relerr(y.cpu().numpy(), yK) <= TOL
    for kw in (dict(detJ=detJ_xs, perm=eperm), dict(perm=eperm)):
        for fl in (F, F | _lib.WF_FLAG_MASS_ELEMENTWISE):
            y = dev(np.zeros(om.ndofs), gpu)
            w.MassOperatorLumped(V, p, structured=False, flags=fl, **kw)(dev(x, gpu), y)
            assert relerr(y.cpu().numpy(), yM) <= 1e-14
    # no perm: the dofmap itself is in x-slowest tensor order
    V2 = w.FunctionSpace(mesh, p, np.ascontiguousarray(dm_xs), w.IndexMap(om.ndofs), om.lattice, structured=False)
    y = dev(np.zeros(om.ndofs), gpu)
    w.StiffnessOperator(V2, p, {"c0": 1500.0}, G=G_xs, structured=False, flags=F)(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yK) <= TOL
    # the hazard: the same x-slowest inputs without the flag give a different operator
    y = dev(np.zeros(om.ndofs), gpu)
    w.StiffnessOperator(V2, p, {"c0": 1500.0}, structured=False)(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yK) > 1e-3


def test_gather_scatter_transform(gpu, oracle):
    """demo/gpu_scatter_local/main.cpp:70,84-90: gather(iota) == dofmap, exactly."""
    import torch
    import wave_fenics_amd as w
    p, n = 3, (4, 3, 2)
    om, mesh, V = make(oracle, n, p)
    Ne = om.dofmap.size
    idx = dev(om.dofmap.reshape(-1), gpu)
    x = torch.arange(om.ndofs, dtype=torch.float64, device=gpu)
    xe = torch.zeros(Ne, dtype=torch.float64, device=gpu)
    w.gather(Ne, idx, x, xe, 512)
    assert np.array_equal(xe.cpu().numpy(), om.dofmap.reshape(-1).astype(np.float64))
    # scatter-add of integer-valued data is exact in any order
    y = torch.zeros(om.ndofs, dtype=torch.float64, device=gpu)
    w.scatter(Ne, idx, xe, y, 512)
    ref = np.zeros(om.ndofs)
    np.add.at(ref, om.dofmap.reshape(-1), om.dofmap.reshape(-1).astype(np.float64))
    assert np.array_equal(y.cpu().numpy(), ref)
    d = dev(np.random.default_rng(0).uniform(0.5, 2, Ne), gpu)
    out = torch.zeros_like(xe)
    w.transform1(Ne, xe, d, out, 512)
    assert np.array_equal(out.cpu().numpy(), xe.cpu().numpy() * d.cpu().numpy())
    # empty input
    w.gather(0, idx, x, xe)
    w.scatter(0, idx, xe, y)


def test_la_kernels(gpu):
    import torch
    from wave_fenics_amd import la
    rng = np.random.default_rng(3)
    n, nl = 100003, 90001
    a, b = rng.uniform(-1, 1, n), rng.uniform(0.5, 2, n)
    x, y = dev(a, gpu), dev(b, gpu)
    r = torch.zeros_like(x)
    la.axpy(r, 0.37, x, y, nl)
    ref = np.zeros(n)
    ref[:nl] = a[:nl] * 0.37 + b[:nl]
    # the device contracts x*alpha + y into one fma (as gcc -Ofast does for the
    # reference's lambda, LinearGLL.hpp:32): equal to one rounding of the product
    ulp = lambda got, want: np.abs(got - want).max() <= 2.3e-16 * np.abs(want).max()
    assert ulp(r.cpu().numpy(), ref) and np.all(r.cpu().numpy()[nl:] == 0.0)
    la.axpy(y, 0.5, x, y, nl)                      # aliasing r == y as LinearGLL.hpp:253
    ref2 = b.copy()
    ref2[:nl] = a[:nl] * 0.5 + b[:nl]
    assert ulp(y.cpu().numpy(), ref2)
    ref2 = y.cpu().numpy()
    out = torch.zeros_like(x)
    la.pointwise_div(x, y, out)
    assert ulp(out.cpu().numpy(), a / ref2)
    la.fill(out, 2.5)
    assert np.all(out.cpu().numpy() == 2.5)
    la.copy(x, out)
    assert np.array_equal(out.cpu().numpy(), a)
    assert abs(la.inner_product(x, y, nl) - np.dot(a[:nl], ref2[:nl])) <= 1e-12 * nl
    la.scale(2.0, out)
    assert np.array_equal(out.cpu().numpy(), 2 * a)


def test_error_behaviour(gpu, oracle):
    import wave_fenics_amd as w
    om, mesh, V = make(oracle, (2, 2, 2), 2)
    with pytest.raises(w.WavehipError):
        w.StiffnessOperator(V, 8, structured=False)             # unsupported degree (mass.hpp:91-92)
    bad = V.dofmap.copy()
    bad[0, 0] = om.ndofs                                          # out-of-range dof index
    Vb = w.FunctionSpace(mesh, 2, bad, w.IndexMap(om.ndofs), V.lattice, structured=False)
    with pytest.raises(w.WavehipError):
        w.StiffnessOperator(Vb, 2)
    import torch
    op = w.StiffnessOperator(V, 2)
    short = torch.zeros(5, dtype=torch.float64, device=gpu)
    with pytest.raises(w.WavehipError):
        op(short, short)


def test_full_size_properties(gpu):
    """BASELINE cfg2 (P4, 54^3 cells, 10 218 313 dofs): size-independent
    properties at full size -- K 1 = 0, x^T K x = -c0^2 |Omega| for x = X,
    symmetry u^T K v = v^T K u, linearity, generic == structured."""
    import torch
    import wave_fenics_amd as w
    p, N = 4, 54
    mesh = w.create_box(N, perturb=0.0)
    V = w.create_functionspace(mesh, p)
    op = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=True)
    n = V.ndofs
    assert n == 10218313
    c02 = 1500.0 ** 2
    y = torch.zeros(n, dtype=torch.float64, device=gpu)
    op(torch.ones(n, dtype=torch.float64, device=gpu), y)
    assert float(y.abs().max()) <= 1e-10 * c02
    NX = V.lattice[0]
    pts, _, _ = w.tabulate_gll(p)
    xs = np.concatenate([(np.arange(N)[:, None] + pts[None, :p]).reshape(-1), [float(N)]]) / N
    X = torch.from_numpy(xs).to(gpu).repeat(V.lattice[1] * V.lattice[2])
    y.zero_()
    op(X, y)
    assert abs(float(torch.dot(X, y)) / (-c02) - 1.0) <= 1e-11
    g = torch.Generator(device=gpu).manual_seed(1)
    u = torch.rand(n, dtype=torch.float64, device=gpu, generator=g) - 0.5
    v = torch.rand(n, dtype=torch.float64, device=gpu, generator=g) - 0.5
    Ku, Kv, Kuv = torch.zeros_like(u), torch.zeros_like(u), torch.zeros_like(u)
    op(u, Ku)
    op(v, Kv)
    a, b = float(torch.dot(v, Ku)), float(torch.dot(u, Kv))
    assert abs(a - b) <= 1e-11 * abs(a)
    op(2.0 * u - 3.0 * v, Kuv)
    assert float((Kuv - (2.0 * Ku - 3.0 * Kv)).abs().max()) <= 1e-11 * float(Kuv.abs().max())
    # generic kernel on the same mesh through the explicit dofmap
    opg = w.StiffnessOperator(V, p, {"c0": 1500.0}, structured=False)
    Kg = torch.zeros_like(u)
    opg(u, Kg)
    assert float((Kg - Ku).abs().max()) <= 1e-11 * float(Ku.abs().max())


def test_rk4_cfg1_vs_oracle(gpu, oracle):
    """BASELINE cfg1: P2 box, 18^3 cells (50 653 dofs), 100 RK4 steps from rest,
    Gamma_1 = face x=0, Gamma_2 = the rest; tolerance 1e-9 relative on u and v
    (SURVEY 8c).  CFL 0.25, see tests/test_oracle_kat.py::test_rk4_runs_and_is_stable."""
    import wave_fenics_amd as w
    from wave_fenics_amd.linear_gll import LinearGLLOpt, cfl_time_step
    p, N = 2, 18
    hi = (0.01, 0.01, 0.01)
    om = oracle.create_box(N, p, hi=hi)
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    dt, _ = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.25)
    ref.init()
    ref.rk4(0.0, 100 * dt - 1e-13, dt)
    mesh = w.create_box(N, hi=hi)
    V = w.create_functionspace(mesh, p)
    dt2, _ = cfl_time_step(mesh, p, 1500.0, 0.5e6, CFL=0.25)
    assert dt2 == dt
    for structured in (True, False):
        eqn = LinearGLLOpt(V, p, 1500.0, 0.5e6, 6e4, structured=structured)
        eqn.init()
        t, steps = eqn.rk4(0.0, 100 * dt - 1e-13, dt)
        assert steps == 100
        assert relerr(eqn.u_n.cpu().numpy(), ref.u_n) <= 1e-9
        assert relerr(eqn.v_n.cpu().numpy(), ref.v_n) <= 1e-9
        # fused stage pipeline (wf_rk4_stage): same expressions, one pass between applies
        fus = LinearGLLOpt(V, p, 1500.0, 0.5e6, 6e4, structured=structured)
        fus.init()
        t, steps = fus.rk4_fused(0.0, 100 * dt - 1e-13, dt)
        assert steps == 100
        assert relerr(fus.u_n.cpu().numpy(), ref.u_n) <= 1e-9
        assert relerr(fus.v_n.cpu().numpy(), ref.v_n) <= 1e-9
        assert relerr(fus.u_n.cpu().numpy(), eqn.u_n.cpu().numpy()) <= 1e-11


def test_rk4_cfg1_reference_cfl(gpu, oracle):
    """BASELINE cfg1 at the REFERENCE's CFL 0.5 (demo/cpu_planar3d/main.cpp:61), 20
    steps.  The other RK4 tests run at CFL 0.25 because 0.5 is unstable on the cube
    (the corner dofs carry three absorbing faces) -- parity does
    not need stability: both sides follow the same (growing) trajectory.  Mesh, time
    step and facet masses come from the oracle and are handed to the product, so
    nothing on the product side is compared with its own twin."""
    import wave_fenics_amd as w
    from wave_fenics_amd.linear_gll import LinearGLLOpt
    p, N = 2, 18
    hi = (0.01, 0.01, 0.01)
    om = oracle.create_box(N, p, hi=hi)
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    dt, _ = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.5)
    ref.init()
    ref.rk4(0.0, 20 * dt - 1e-13, dt)
    mesh = w.BoxMesh(om.n, om.x.copy(), om.geom_dofmap.copy(), (0.0, 0.0, 0.0), hi)
    V = w.create_functionspace(mesh, p)
    assert np.array_equal(V.dofmap, om.dofmap)
    sets = []
    for tag in (1, 2):
        dense = oracle.facet_lumped_mass(om, tag)
        idx = np.nonzero(dense)[0].astype(np.int32)
        sets.append((idx, dense[idx]))
    for structured in (True, False):
        for fused in (False, True):
            eqn = LinearGLLOpt(V, p, 1500.0, 0.5e6, 6e4, boundary=tuple(sets), structured=structured)
            eqn.init()
            t, steps = (eqn.rk4_fused if fused else eqn.rk4)(0.0, 20 * dt - 1e-13, dt)
            assert steps == 20
            assert relerr(eqn.u_n.cpu().numpy(), ref.u_n) <= 1e-9
            assert relerr(eqn.v_n.cpu().numpy(), ref.v_n) <= 1e-9


def _dist_gpu_worker(rank, world, port, n, p, q):
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import torch
        import torch.distributed as dist
        from dist_helpers import init_pg, local_to_global
        from oracle import wave_oracle as o
        import wave_fenics_amd as w
        from wave_fenics_amd.distributed import VectorUpdater, boundary_tags, create_distributed_box
        from wave_fenics_amd.linear_gll import LinearGLLOpt
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        init_pg(rank, world, port, "gloo")      # one GPU box: ranks share cuda:0, transport staged through the host
        hi = (0.01, 0.01, 0.01)
        part = create_distributed_box(n, p, world, rank, hi=hi, build_dofmap=True)
        vu = VectorUpdater(part, device=dev)
        l2g = local_to_global(part)
        owned = part.owned_mask()
        gn = tuple(part.procs[a] * n[a] for a in range(3))
        gm = o.create_box(gn, p, hi=hi)
        errs = {}
        for structured in (True, False):
            V = part.V
            V.structured = structured
            K = w.StiffnessOperator(V, p, {"c0": 1500.0})
            xg = np.random.default_rng(5).uniform(-1, 1, gm.ndofs)
            xl = np.zeros(l2g.size)
            xl[owned] = xg[l2g[owned]]
            x = torch.from_numpy(xl).to(dev)
            y = torch.zeros_like(x)
            vu.update_fwd(x)
            K(x, y)
            vu.update_rev(y)
            yg = np.zeros(gm.ndofs)
            o.StiffnessOperator(gm, p)(xg, yg)
            errs[f"K{int(structured)}"] = float(np.abs(y.cpu().numpy()[owned] - yg[l2g[owned]]).max() / np.abs(yg).max())
        # full RK4 loop with ghost exchange (LinearGLL.hpp:164-176) vs the single-domain oracle
        part.V.structured = True
        ref = o.LinearGLLOpt(gm, p, 1500.0, 0.5e6, 6e4)
        dt, _ = o.cfl_time_step(gm, p, 1500.0, 0.5e6, CFL=0.25)
        ref.init()
        ref.rk4(0.0, 20 * dt - 1e-13, dt)
        eqn = LinearGLLOpt(part.V, p, 1500.0, 0.5e6, 6e4, updater=vu, tags=boundary_tags(part), device=dev)
        eqn.init()
        eqn.rk4(0.0, 20 * dt - 1e-13, dt)
        u = eqn.u_n.cpu().numpy()
        errs["rk4_u"] = float(np.abs(u - ref.u_n[l2g]).max() / np.abs(ref.u_n).max())    # ghosts included (final scatter_fwd)
        errs["rk4_v"] = float(np.abs(eqn.v_n.cpu().numpy() - ref.v_n[l2g]).max() / np.abs(ref.v_n).max())
        fus = LinearGLLOpt(part.V, p, 1500.0, 0.5e6, 6e4, updater=vu, tags=boundary_tags(part), device=dev)
        fus.init()
        fus.rk4_fused(0.0, 20 * dt - 1e-13, dt)
        errs["rk4f_u"] = float(np.abs(fus.u_n.cpu().numpy() - ref.u_n[l2g]).max() / np.abs(ref.u_n).max())
        errs["rk4f_v"] = float(np.abs(fus.v_n.cpu().numpy() - ref.v_n[l2g]).max() / np.abs(ref.v_n).max())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, errs, None))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.parametrize("world,n,p", [(2, (3, 4, 4), 2), (4, (3, 3, 4), 4)])
def test_distributed_on_one_gpu(gpu, oracle, world, n, p):
    """Domain-decomposed apply and RK4 loop with the real HIP pack/unpack kernels:
    `world` ranks share cuda:0, torch.distributed gloo carries the halo (staged
    through the host).  The 8-GPU RCCL run uses the same code with device buffers."""
    import torch.multiprocessing as mp
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, sys_path)
    from dist_helpers import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_dist_gpu_worker, args=(r, world, port, n, p, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
    for rank, errs, tb in res:
        assert tb is None, f"rank {rank} failed:\n{tb}"
        assert errs["K1"] <= 1e-12 and errs["K0"] <= 1e-12, errs
        assert errs["rk4_u"] <= 1e-9 and errs["rk4_v"] <= 1e-9, errs
        assert errs["rk4f_u"] <= 1e-9 and errs["rk4f_v"] <= 1e-9, errs


@pytest.mark.parametrize("p,n,perturb", [(1, (3, 3, 2), 0.2), (2, (3, 2, 2), 0.2), (3, (2, 2, 2), 0.2), (4, (3, 2, 2), 0.2),
                                         (4, (2, 2, 1), 0.0), (4, (5, 3, 1), 0.15)])
def test_tet_dense_stiffness_vs_oracle(gpu, oracle, p, n, perturb):
    """BASELINE configs[4] at small size: the dense MFMA path on Kuhn tetrahedra
    against the dense skernel oracle with tetrahedral tables (parity unpinned,
    oracle/tet_oracle.py).  Partial batches (ncells not a multiple of 64) included."""
    from oracle import tet_oracle
    from wave_fenics_amd import tet
    om = tet_oracle.create_kuhn_box(n, p, perturb=perturb)
    V = tet.create_kuhn_box(n, p, perturb=perturb)
    assert np.array_equal(V.dofmap, om.dofmap) and np.array_equal(V.geom_dofmap, om.geom_dofmap)
    K = tet_oracle.TetStiffnessOperator(om, p)
    rng = np.random.default_rng(4242)
    x = rng.uniform(-1, 1, om.ndofs)
    y0 = rng.uniform(-1, 1, om.ndofs) * 1e6
    yref = y0.copy()
    K(x, yref)
    op = tet.TetStiffnessOperator(V, p, {"c0": 1500.0})
    y = dev(y0, gpu)
    op(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= 1e-12
    assert op.num_dofs() == K.nd and op.num_quads() == K.nq


def test_full_size_cfg3_p6_mass(gpu):
    """BASELINE configs[2] at full size (P6, 36^3 cells, 10 218 313 dofs): lumped
    (spectral) mass in both forms and the dense Phi^T D Phi form (collocated rule)
    agree; sum(M 1) = |Omega|; M is positive."""
    import torch
    import wave_fenics_amd as w
    p, N = 6, 36
    mesh = w.create_box(N, perturb=0.0)
    V = w.create_functionspace(mesh, p)
    n = V.ndofs
    assert n == 10218313
    ones = torch.ones(n, dtype=torch.float64, device=gpu)
    m_box = torch.zeros_like(ones)
    w.SpectralMassOperator(V, p, structured=True).apply(ones, m_box)
    m_gen = torch.zeros_like(ones)
    w.SpectralMassOperator(V, p, structured=False).apply(ones, m_gen)
    assert abs(float(m_box.sum()) - 1.0) <= 1e-12 and float(m_box.min()) > 0
    assert float((m_box - m_gen).abs().max()) <= 1e-13 * float(m_box.max())
    pts, wts, _ = w.tabulate_gll(p)
    W3 = np.einsum("k,j,i->kji", wts, wts, wts).reshape(-1)
    detJ = np.tile(W3 / mesh.ncells, (mesh.ncells, 1))
    m_dense = torch.zeros_like(ones)
    w.MassOperator(V, p, np.eye(p + 1), detJ).apply(ones, m_dense)
    assert float((m_dense - m_box).abs().max()) <= 1e-13 * float(m_box.max())
    g = torch.Generator(device=gpu).manual_seed(2)
    x = torch.rand(n, dtype=torch.float64, device=gpu, generator=g)
    y = torch.zeros_like(x)
    w.SpectralMassOperator(V, p, structured=True).apply(x, y)
    assert float((y - m_box * x).abs().max()) <= 1e-15


def test_full_size_cfg5_tets(gpu):
    """BASELINE configs[4] at full size (P4 tetrahedra, Kuhn split of 54^3 cubes,
    944 784 cells, 10 218 313 dofs): K 1 = 0, X^T K X = -c0^2 |Omega|, symmetry."""
    import torch
    from wave_fenics_amd import tet
    p, N = 4, 54
    V = tet.create_kuhn_box(N, p)
    assert V.ncells == 944784 and V.ndofs == 10218313
    op = tet.TetStiffnessOperator(V, p, {"c0": 1500.0})
    n = V.ndofs
    c02 = 1500.0 ** 2
    y = torch.zeros(n, dtype=torch.float64, device=gpu)
    op(torch.ones(n, dtype=torch.float64, device=gpu), y)
    assert float(y.abs().max()) <= 1e-9 * c02
    NX = V.lattice[0]
    X = (torch.arange(NX, dtype=torch.float64, device=gpu) / (NX - 1)).repeat(V.lattice[1] * V.lattice[2])
    y.zero_()
    op(X, y)
    assert abs(float(torch.dot(X, y)) / (-c02) - 1.0) <= 1e-10
    g = torch.Generator(device=gpu).manual_seed(3)
    u = torch.rand(n, dtype=torch.float64, device=gpu, generator=g) - 0.5
    v = torch.rand(n, dtype=torch.float64, device=gpu, generator=g) - 0.5
    Ku, Kv = torch.zeros_like(u), torch.zeros_like(u)
    op(u, Ku)
    op(v, Kv)
    a, b = float(torch.dot(v, Ku)), float(torch.dot(u, Kv))
    assert abs(a - b) <= 1e-10 * abs(a)


def test_cxx_host_driver_cfg1(gpu, oracle, tmp_path):
    """The C++ host side (include/wavehip_linear_gll.hpp: class LinearGLLOpt with
    the reference's members and call order, examples/planar3d.cpp = the reference's
    demo/cpu_planar3d/main.cpp) on BASELINE cfg1: P2, 18^3 cells, 100 RK4 steps,
    against the oracle to 1e-9; plus the operator demo's printed norms."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "bin")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), f"OUT={out}"])
    p, N = 2, 18
    hi = (0.01, 0.01, 0.01)
    om = oracle.create_box(N, p, hi=hi)
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    dt, spp = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.25)
    ref.init()
    ref.rk4(0.0, 100 * dt - 1e-13, dt)
    dump = str(tmp_path / "uv.bin")
    r = subprocess.run([os.path.join(out, "planar3d"), "--size", str(N), "--degree", str(p), "--cfl", "0.25",
                        "--steps", "100", "--length", "0.01", "--dump", dump], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"Number of step per period: {spp}" in r.stdout
    assert "Steps taken: 100" in r.stdout and f"Degrees of freedom: {om.ndofs}" in r.stdout
    uv = np.fromfile(dump, dtype=np.float64)
    assert uv.size == 2 * om.ndofs
    assert relerr(uv[: om.ndofs], ref.u_n) <= 1e-9
    assert relerr(uv[om.ndofs:], ref.v_n) <= 1e-9
    # operator demo: lumped mass with x = 1 (the reference's --check input)
    om4 = oracle.create_box(6, 4)
    m = np.zeros(om4.ndofs)
    oracle.MassOperatorCPU(om4, 4)(np.ones(om4.ndofs), m)
    r = subprocess.run([os.path.join(out, "operator_demo"), "--size", "6", "--degree", "4", "--op", "mass", "--reps", "2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    vals = {l.split(":")[0].strip(): l.split(":")[1].strip() for l in r.stdout.splitlines() if ":" in l}
    assert abs(float(vals["Y norm"]) - np.linalg.norm(m)) <= 1e-5 * np.linalg.norm(m)     # printed with 6 digits
    assert int(vals["Number of cells"]) == 216 and int(vals["Number of dofs"]) == 125
    # --op dense --check (gpu_operator_monolithic/main.cpp:93-118): the dense MassOperator built from
    # the reference's element / quadrature arguments against the lumped operator with x = 1
    r = subprocess.run([os.path.join(out, "operator_demo"), "--size", "5", "--degree", "3", "--op", "dense", "--check",
                        "--reps", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "entries differing by > 1e-8: 0" in r.stdout and "Number of quads: 64" in r.stdout
    r = subprocess.run([os.path.join(out, "operator_demo"), "--size", "4", "--degree", "2", "--op", "dense", "--check",
                        "--variant", "equispaced", "--quad", "gauss", "--qdegree", "4", "--reps", "2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("check:")][0].split()
    assert abs(float(line[3]) - 1.0) <= 1e-12 and abs(float(line[6]) - 1.0) <= 1e-12      # both sum to the volume


def test_generic_ops_random_cell_order(gpu, oracle):
    """Arbitrary cell order (the operators sort cells internally by their smallest
    dof) with every way of handing over the per-cell data: G / detJ arrays in the
    caller's cell order, or the mesh."""
    import wave_fenics_amd as w
    p, n = 3, (4, 3, 3)
    om, mesh, V = make(oracle, n, p)
    rng = np.random.default_rng(21)
    cp = rng.permutation(om.ncells)
    gperm = rng.permutation(om.ndofs).astype(np.int32)
    dm = gperm[om.dofmap][cp]
    m2 = w.BoxMesh(mesh.n, mesh.x, np.ascontiguousarray(mesh.geom_dofmap[cp]))
    V2 = w.FunctionSpace(m2, p, np.ascontiguousarray(dm), w.IndexMap(om.ndofs), V.lattice, structured=False)
    K = oracle.StiffnessOperator(om, p)
    M = oracle.MassOperatorCPU(om, p)
    x = rng.uniform(-1, 1, om.ndofs)
    xp = np.empty_like(x)
    xp[gperm] = x
    yK, yM = np.zeros(om.ndofs), np.zeros(om.ndofs)
    K(x, yK)
    M(x, yM)
    for kw in (dict(G=K.G[cp]), dict()):
        y = dev(np.zeros(om.ndofs), gpu)
        w.StiffnessOperator(V2, p, {"c0": 1500.0}, structured=False, **kw)(dev(xp, gpu), y)
        assert relerr(y.cpu().numpy()[gperm], yK) <= 1e-12
    for kw in (dict(detJ=M.detJ[cp]), dict()):
        y = dev(np.zeros(om.ndofs), gpu)
        w.MassOperatorLumped(V2, p, structured=False, **kw)(dev(xp, gpu), y)
        assert relerr(y.cpu().numpy()[gperm], yM) <= 1e-13
    pts, wts, phi1, phi, X, W = oracle.tabulate_mass_tables(p, "equispaced", "gauss_jacobi", 2 * p)
    detq = oracle.compute_detJ_generic(om, X, W)
    yD = np.zeros(om.ndofs)
    oracle.dense_mass_apply(om, phi, detq, x, yD)
    y = dev(np.zeros(om.ndofs), gpu)
    w.MassOperator(V2, p, phi1, detq[cp]).apply(dev(xp, gpu), y)
    assert relerr(y.cpu().numpy()[gperm], yD) <= 1e-12


@pytest.mark.parametrize("ncells,K,N", [(1000, 125, 125), (37, 27, 27), (513, 64, 64), (100, 8, 8), (77, 35, 192),
                                        (50, 125, 216), (16, 3, 5), (300, 216, 216), (130, 343, 343), (40, 512, 512),
                                        (64, 129, 40), (33, 216, 125),
                                        # whole rounds of the persistent grid (2048 waves x 16 cells) + a last round that is
                                        # split along the columns into units of 1 / 2 column tiles, or too full to split
                                        (100000, 125, 125), (40000, 125, 125), (60000, 125, 125), (70000, 27, 27),
                                        (35000, 216, 216), (32768, 64, 64)])
def test_tsmm_vs_numpy(gpu, ncells, K, N):
    """wf_tsmm in both array layouts (demo/gpu_operator cell-major, demo/gpu_tsmm
    column-major with lda = ncells) against a float64 numpy product: k-ordered
    fma chains, tolerance 1e-13 of the row magnitude.  K > 128 (the P5..P7 tables of
    demo/gpu_operator: 216, 343, 512) runs as row ranges accumulating onto out."""
    import torch
    import wave_fenics_amd as w
    rng = np.random.default_rng(ncells + K)
    A = rng.uniform(-1, 1, (ncells, K))
    B = rng.uniform(-1, 1, (K, N))
    C = A @ B
    scale = (np.abs(A) @ np.abs(B)).max()
    out = torch.zeros(ncells * N, dtype=torch.float64, device=gpu)
    w.tsmm(ncells, dev(A.reshape(-1), gpu), dev(B, gpu), out, layout=0)
    assert np.abs(out.cpu().numpy().reshape(ncells, N) - C).max() <= 1e-13 * scale
    out.zero_()
    w.tsmm(ncells, dev(np.ascontiguousarray(A.T).reshape(-1), gpu), dev(B, gpu), out, layout=1)
    assert np.abs(out.cpu().numpy().reshape(N, ncells).T - C).max() <= 1e-13 * scale


def test_edge_cases_single_cell_and_empty(gpu, oracle):
    """Smallest inputs: one cell (every kernel's partial-batch path), an operator
    with zero local cells (a rank that owns no cell), zero-length vector kernels."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd import la
    for p in (1, 2, 4, 7):
        om, mesh, V = make(oracle, (1, 1, 1), p, perturb=0.0)
        K = oracle.StiffnessOperator(om, p)
        M = oracle.MassOperatorCPU(om, p)
        x = np.random.default_rng(p).uniform(-1, 1, om.ndofs)
        yK, yM = np.zeros(om.ndofs), np.zeros(om.ndofs)
        K(x, yK)
        M(x, yM)
        for structured in (True, False):
            y = dev(np.zeros(om.ndofs), gpu)
            w.StiffnessOperator(V, p, structured=structured)(dev(x, gpu), y)
            assert relerr(y.cpu().numpy(), yK) <= 1e-12
            y = dev(np.zeros(om.ndofs), gpu)
            w.MassOperatorLumped(V, p, structured=structured)(dev(x, gpu), y)
            assert relerr(y.cpu().numpy(), yM) <= 1e-13
    # zero local cells
    mesh0 = w.BoxMesh((0, 0, 0), np.zeros((1, 3)), np.zeros((0, 8), dtype=np.int32))
    V0 = w.FunctionSpace(mesh0, 2, np.zeros((0, 27), dtype=np.int32), w.IndexMap(5), (0, 0, 0), structured=False)
    x = torch.ones(5, dtype=torch.float64, device=gpu)
    y = torch.full((5,), 3.0, dtype=torch.float64, device=gpu)
    for op in (w.StiffnessOperator(V0, 2, structured=False), w.MassOperatorLumped(V0, 2, structured=False)):
        op(x, y)
        assert op.num_cells() == 0
    assert bool((y == 3.0).all())
    # zero-length vector kernels are no-ops
    e = torch.zeros(0, dtype=torch.float64, device=gpu)
    la.axpy(e, 1.0, e, e)
    la.fill(e, 1.0)
    la.copy(e, e)
    la.pointwise_div(e, e, e)
    assert la.inner_product(e, e) == 0.0


def test_mass_with_element_permutation(gpu, oracle):
    """SpectralMassOperator's reorder_dofmap path (common/cuda/spectral_mass.hpp:50-53,
    common/permute.hpp:10-27): element-ordered dofmap + perm."""
    import wave_fenics_amd as w
    p, n = 3, (3, 2, 2)
    om, mesh, V = make(oracle, n, p)
    rng = np.random.default_rng(17)
    eperm = rng.permutation((p + 1) ** 3).astype(np.int32)
    inv = np.empty_like(eperm)
    inv[eperm] = np.arange(eperm.size, dtype=np.int32)
    dm_elem = om.dofmap[:, inv]
    Vp = w.FunctionSpace(mesh, p, np.ascontiguousarray(dm_elem), w.IndexMap(om.ndofs), V.lattice, structured=False)
    M = oracle.MassOperatorCPU(om, p)
    x = rng.uniform(-1, 1, om.ndofs)
    yref = np.zeros(om.ndofs)
    M(x, yref)
    y = dev(np.zeros(om.ndofs), gpu)
    w.MassOperatorLumped(Vp, p, perm=eperm)(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= 1e-13


@pytest.mark.parametrize("p,n", [(1, (70, 33, 21)), (2, (41, 37, 19)), (3, (23, 29, 17)), (4, (31, 13, 29)),
                                 (4, (54, 1, 1)), (4, (1, 1, 40)), (5, (11, 12, 13)), (6, (9, 10, 7)), (7, (6, 7, 5))])
def test_box_vs_generic_medium_meshes(gpu, p, n):
    """Medium, non-cubic, perturbed meshes (default segmentation, partial columns and
    segments): the structured operator and the generic operator (explicit dofmap,
    batch-unique scatter) agree to 1e-11 -- two independent gather/scatter paths
    around the same element kernel, at sizes the CPU oracle would not finish."""
    import torch
    import wave_fenics_amd as w
    mesh = w.create_box(n, perturb=0.2)
    V = w.create_functionspace(mesh, p)
    g = torch.Generator(device=gpu).manual_seed(11)
    x = torch.rand(V.ndofs, dtype=torch.float64, device=gpu, generator=g) - 0.5
    ys = torch.zeros_like(x)
    yg = torch.zeros_like(x)
    w.StiffnessOperator(V, p, structured=True)(x, ys)
    w.StiffnessOperator(V, p, structured=False)(x, yg)
    assert float((ys - yg).abs().max()) <= 1e-11 * float(yg.abs().max())
    # K 1 = 0 on the perturbed mesh (up to the clamp of tiny G entries)
    y1 = torch.zeros_like(x)
    w.StiffnessOperator(V, p, structured=True)(torch.ones_like(x), y1)
    assert float(y1.abs().max()) <= 1e-6 * float(yg.abs().max())
