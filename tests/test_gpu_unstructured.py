"""GPU parity on meshes that are NOT a consistently oriented box -- what a gmsh / multi-block
hexahedral mesh (such as the reference's absent mesh.xdmf, demo/cpu_planar3d/main.cpp:39-45)
looks like -- and on every kernel an operator can end up on:

  * two box blocks glued together with the second block's cells locally rotated / reflected,
    a single flipped cell, every cell in a random one of the 48 orientations, and an O-grid
    (three blocks around an irregular edge);
  * the lattice-column marching kernels with orientation normalisation (k_march_idx,
    k_march_ks), the batch kernels a mesh that does not tile gets (k_stiffness_generic_u,
    k_mass_lumped_u, k_mass_dense_col, k_mass_dense) and the element-wise forms, each selected
    explicitly through wf_tuning and CONFIRMED through wf_op_info_t.kernel;
  * the native RCCL exchange with several segments to one peer (the 7-neighbour code path of
    cfg4: recvbuf + roff[i], sendbuf + soff[i], i > 0, a zero-length segment, several sends
    in one group), through the C ABI and through wavehip::VectorUpdater;
  * the file rendezvous with a polling rank.

Reference semantics: StiffnessOperator::operator() (common/operators.hpp:183-200), mkernel
(operators.hpp:36-40), MassOperator::apply (common/cuda/mass.hpp:76-95), VectorUpdater
(demo/gpu_scatter_mpi/VectorUpdater.hpp:106-208).  Tolerance: one apply 1e-12 (fp64, summation
order); index work bit-exact."""
import ctypes
import os
import subprocess
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import wave_fenics_amd as w
    w.lib()
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def dev(a, gpu):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def build_mesh(kind: str, p: int):
    """(mesh, expected number of cells the lattice plan must re-orient, or None)"""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    if kind == "glued_rotated":      # block 2 (x >= 3 cells) rotated by 90 degrees about z: new x = old y, new y = -old x
        box = w.create_box((6, 4, 3), perturb=0.2)
        sel = np.nonzero(np.arange(box.ncells) % 6 >= 3)[0]
        return mesh_io.reorient_cells(box, sel, 8 * 2 + 2), len(sel)
    if kind == "glued_reflected":    # block 2 mirrored in its local z and with x, z exchanged
        box = w.create_box((5, 4, 4), perturb=0.2)
        sel = np.nonzero(np.arange(box.ncells) % 5 >= 2)[0]
        return mesh_io.reorient_cells(box, sel, 8 * 5 + 4), len(sel)
    if kind == "one_flipped":
        box = w.create_box((5, 4, 3), perturb=0.2)
        return mesh_io.reorient_cells(box, [17], 8 * 3 + 1), 1
    if kind == "random_orient":
        box = w.create_box((5, 4, 3), perturb=0.2)
        codes = np.random.default_rng(5).integers(0, 48, box.ncells)
        codes[0] = 0                 # the seed cell's frame is the lattice frame
        return mesh_io.reorient_cells(box, np.arange(box.ncells), codes), None
    if kind == "ogrid":
        return mesh_io.create_ogrid(3 if p <= 4 else 2, 3, perturb=0.15), None
    raise ValueError(kind)


def oracle_mesh(oracle, mesh, V):
    return oracle.BoxMesh(None, V.degree, np.ascontiguousarray(mesh.x), np.ascontiguousarray(mesh.geom_dofmap),
                          np.ascontiguousarray(V.dofmap), V.ndofs, None)


MESHES = ["glued_rotated", "glued_reflected", "one_flipped", "random_orient", "ogrid"]


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("kind", MESHES)
def test_stiffness_on_reoriented_meshes(gpu, oracle, kind, p):
    """y += K x on meshes whose cells do not agree on their local axes, for the lattice plan with
    orientation normalisation (default), the plan forced whatever its fill, the batch kernel and
    the element-wise kernel; geometry computed on the device from the mesh and handed over in
    the reference layout (the 3 x 3 axes of G are relabelled with the cell)."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    mesh, nre = build_mesh(kind, p)
    V = mesh_io.create_functionspace(mesh, p)
    om = oracle_mesh(oracle, mesh, V)
    K = oracle.StiffnessOperator(om, p)
    rng = np.random.default_rng(1234)
    x = rng.uniform(-1, 1, V.ndofs)
    y0 = rng.uniform(-1, 1, V.ndofs) * 1e6          # accumulate semantics: y += K x
    yref = y0.copy()
    K(x, yref)
    seen = set()
    for hint, Garg in (("auto", None), ("march", None), ("march", K.G), ("batch", None), ("batch", K.G), ("elementwise", None)):
        op = w.StiffnessOperator(V, p, {"c0": 1500.0}, G=Garg, structured=False, tuning={"kernel": hint})
        seen.add(op.kernel)
        if hint == "march":
            assert op.kernel == "march_idx", (kind, p, op.kernel)
            assert 0.0 < op.info.plan_fill <= 1.0 and op.info.plan_items > 0
            if nre is not None:
                assert op.info.plan_reoriented == nre
            if kind == "random_orient":
                assert op.info.plan_reoriented > mesh.ncells // 2
        elif hint == "batch":
            assert op.kernel == "batch_unique"
        elif hint == "elementwise":
            assert op.kernel == "elementwise"
        else:
            assert op.kernel in ("march_idx", "batch_unique")
        y = dev(y0, gpu)
        op(dev(x, gpu), y)
        err = relerr(y.cpu().numpy(), yref)
        assert err <= TOL, (kind, p, hint, Garg is not None, op.kernel, err)
    assert {"march_idx", "batch_unique", "elementwise"} <= seen


def test_plan_requires_agreeing_frames_when_told_so(gpu, oracle):
    """wf_tuning.orient = 1 switches the normalisation off: a mesh with one flipped cell then keeps
    only the cells that agree with their neighbours in a column, the flipped one becomes its own
    component -- still the right answer."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    p = 3
    mesh, _ = build_mesh("one_flipped", p)
    V = mesh_io.create_functionspace(mesh, p)
    om = oracle_mesh(oracle, mesh, V)
    K = oracle.StiffnessOperator(om, p)
    x = np.random.default_rng(2).uniform(-1, 1, V.ndofs)
    yref = np.zeros(V.ndofs)
    K(x, yref)
    op = w.StiffnessOperator(V, p, structured=False, tuning={"kernel": "march", "orient": 1})
    assert op.kernel == "march_idx" and op.info.plan_reoriented == 0
    y = dev(np.zeros(V.ndofs), gpu)
    op(dev(x, gpu), y)
    assert relerr(y.cpu().numpy(), yref) <= TOL


@pytest.mark.parametrize("p", [1, 2, 4, 6])
@pytest.mark.parametrize("kind", ["glued_rotated", "glued_reflected", "random_orient", "ogrid"])
def test_mass_operators_on_reoriented_meshes(gpu, oracle, kind, p):
    """Lumped mass (pre-assembled diagonal, the reference's element-wise gather * detJ -> scatter
    in its batch-unique and flat forms) and dense mass Phi^T D Phi (lattice plan, batch column
    kernel, any-rule kernel; GLL-collocated and Gauss rules) on the same meshes."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    from wave_fenics_amd._lib import WF_FLAG_MASS_ELEMENTWISE
    mesh, _ = build_mesh(kind, p)
    V = mesh_io.create_functionspace(mesh, p)
    om = oracle_mesh(oracle, mesh, V)
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, V.ndofs)
    y0 = rng.uniform(-1, 1, V.ndofs)
    # lumped
    M = oracle.MassOperatorCPU(om, p)
    yref = y0.copy()
    M(x, yref)
    for flags, tuning, want in ((0, None, "diagonal"), (WF_FLAG_MASS_ELEMENTWISE, None, "batch_unique"),
                                (WF_FLAG_MASS_ELEMENTWISE, {"kernel": "elementwise"}, "elementwise")):
        op = w.MassOperatorLumped(V, p, structured=False, flags=flags, tuning=tuning)
        assert op.kernel == want
        y = dev(y0, gpu)
        op(dev(x, gpu), y)
        assert relerr(y.cpu().numpy(), yref) <= TOL, (kind, p, want)
    # dense, two rules: collocated GLL (square table) and Gauss of degree 2P (square table, not collocated)
    for variant, quad, qd in (("gll", "gll", {1: 1, 2: 3, 4: 6, 6: 10}[p]), ("equispaced", "gauss_jacobi", 2 * p)):
        pts, wts, phi1, phi, Xq, Wq = oracle.tabulate_mass_tables(p, variant, quad, qd)
        detJ = oracle.compute_detJ_generic(om, Xq, Wq)
        yref = y0.copy()
        oracle.dense_mass_apply(om, phi, detJ, x, yref)
        for hint, want in (("march", "march_idx"), ("batch", "batch_unique"), ("mass_any", "mass_dense_any")):
            op = w.MassOperator(V, p, variant=variant, quad=quad, qdegree=qd, tuning={"kernel": hint})
            assert op.kernel == want, (kind, p, hint, op.kernel)
            y = dev(y0, gpu)
            op(dev(x, gpu), y)
            err = relerr(y.cpu().numpy(), yref)
            assert err <= TOL, (kind, p, variant, hint, err)


def test_thin_column_mesh_keeps_the_batch_kernel(gpu, oracle):
    """A mesh one cell wide fills a fifth of the slots of its lattice columns (P4: 5 x 1 cells per
    layer): the marching kernel would read five times the geometry -- the plan is dropped on its
    fill factor (and adopted when forced, with the same answer)."""
    import wave_fenics_amd as w
    p, n = 4, (1, 1, 24)
    om = oracle.create_box(n, p, perturb=0.0)
    mesh = w.create_box(n)
    V = w.create_functionspace(mesh, p)
    K = oracle.StiffnessOperator(om, p)
    x = np.random.default_rng(3).uniform(-1, 1, om.ndofs)
    yref = np.zeros(om.ndofs)
    K(x, yref)
    op = w.StiffnessOperator(V, p, structured=False)
    assert op.kernel == "batch_unique"
    opm = w.StiffnessOperator(V, p, structured=False, tuning={"kernel": "march"})
    assert opm.kernel == "march_idx" and opm.info.plan_fill <= 0.2
    for o in (op, opm):
        y = dev(np.zeros(om.ndofs), gpu)
        o(dev(x, gpu), y)
        assert relerr(y.cpu().numpy(), yref) <= TOL


# ---------------------------------------------------------------------------------------------
# several segments to one peer
# ---------------------------------------------------------------------------------------------
SEGMENTS = [1201, 0, 37, 1, 640]       # face-, nothing, edge-, corner-, face-sized


def _segment_lists(N, rng):
    total = sum(SEGMENTS)
    perm = rng.permutation(N)[: 2 * total]
    send_idx, ghost_pos = perm[:total].astype(np.int32), perm[total:].astype(np.int32)
    off = np.concatenate([[0], np.cumsum(SEGMENTS)]).astype(np.int32)
    return send_idx, ghost_pos, off


def test_updater_several_segments_to_one_peer(gpu):
    """wf_updater_create with five segments (one empty) that all go to rank 0 -- the self peer of a
    one-rank communicator -- i.e. five ncclSend / ncclRecv pairs in one group with non-zero buffer
    offsets: fwd, rev, the _begin/_end split with work in between, bit-exact against the index
    lists (VectorUpdater.hpp:106-208)."""
    import torch
    from wave_fenics_amd import _lib
    from wave_fenics_amd.comm import Comm
    from wave_fenics_amd.operators import _ptr, _stream
    L = _lib.lib()
    comm = Comm.single()
    N = 20000
    rng = np.random.default_rng(21)
    send_idx, ghost_pos, off = _segment_lists(N, rng)
    nb = np.zeros(len(SEGMENTS), dtype=np.intc)
    d = _lib.UpdaterDesc()
    d.ndofs = N
    d.num_send_neighbors = d.num_recv_neighbors = len(SEGMENTS)
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    d.send_neighbors = d.recv_neighbors = nb.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    d.send_offsets = d.recv_offsets = ip(off)
    d.send_indices, d.ghost_positions = ip(send_idx), ip(ghost_pos)
    for flags in (_lib.WF_UPDATER_DEFAULT, _lib.WF_UPDATER_INLINE):
        d.flags = flags
        h = ctypes.c_void_p()
        _lib.check(L.wf_updater_create(comm._h, ctypes.byref(d), ctypes.byref(h)))
        ns, nr, nsn, nrn = (ctypes.c_int() for _ in range(4))
        _lib.check(L.wf_updater_info(h, ctypes.byref(ns), ctypes.byref(nr), ctypes.byref(nsn), ctypes.byref(nrn)))
        assert (ns.value, nr.value, nsn.value, nrn.value) == (sum(SEGMENTS), sum(SEGMENTS), len(SEGMENTS), len(SEGMENTS))
        x0 = rng.integers(-1000, 1000, N).astype(np.float64)
        # forward: ghost k of segment i <- owner k of segment i
        x = dev(x0, gpu)
        _lib.check(L.wf_updater_fwd(h, _ptr(x), _stream(x)))
        torch.cuda.synchronize()
        want = x0.copy()
        want[ghost_pos] = x0[send_idx]
        assert np.array_equal(x.cpu().numpy(), want)
        # reverse: owner += ghost (small integers: exact)
        y = dev(x0, gpu)
        _lib.check(L.wf_updater_rev(h, _ptr(y), _stream(y)))
        torch.cuda.synchronize()
        want = x0.copy()
        want[send_idx] += x0[ghost_pos]
        assert np.array_equal(y.cpu().numpy(), want)
        # begin / end with independent work between them
        x2 = dev(x0, gpu)
        z = torch.zeros(1 << 20, dtype=torch.float64, device=gpu)
        _lib.check(L.wf_updater_fwd_begin(h, _ptr(x2), _stream(x2)))
        z += 1.0
        _lib.check(L.wf_updater_fwd_end(h, _ptr(x2), _stream(x2)))
        y2 = dev(x0, gpu)
        _lib.check(L.wf_updater_rev_begin(h, _ptr(y2), _stream(y2)))
        z += 1.0
        _lib.check(L.wf_updater_rev_end(h, _ptr(y2), _stream(y2)))
        torch.cuda.synchronize()
        w1 = x0.copy()
        w1[ghost_pos] = x0[send_idx]
        assert np.array_equal(x2.cpu().numpy(), w1) and np.array_equal(y2.cpu().numpy(), want)
        assert float(z.sum()) == float(2 << 20)
        _lib.check(L.wf_updater_destroy(h))
    comm.close()


def test_overlapped_apply_with_segmented_exchange(gpu, oracle):
    """wf_op_apply_overlapped on an xyz-periodic partition whose exchange lists are cut into
    several segments to the self peer (faces / edges / corner as the 2 x 2 x 2 partition of cfg4
    would send them to seven different ranks), for the box operator and the arbitrary-dofmap one,
    against the oracle on the periodic mesh."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd import _lib
    from wave_fenics_amd.comm import Comm
    from wave_fenics_amd.distributed import create_distributed_box
    from wave_fenics_amd.operators import _ptr, _stream
    L = _lib.lib()
    p, n, periodic = 4, (6, 4, 7), (True, True, True)
    part = create_distributed_box(n, p, 1, 0, perturb=0.15, periodic=periodic, build_dofmap=True)
    om = oracle.create_box(n, p, perturb=0.15)
    l2g = oracle.make_periodic(om, periodic)
    owned = part.owned_mask()
    send, recv = part.send_fwd[0], part.recv_fwd[0]
    assert send.size == recv.size and send.size > 100
    # cut the one self segment into 7 pieces of very different sizes (one empty)
    cuts = np.array([0, 1, 1, 40, 41, send.size // 2, send.size - 3, send.size], dtype=np.int32)
    nb = np.zeros(7, dtype=np.intc)
    d = _lib.UpdaterDesc()
    d.ndofs = part.V.ndofs
    d.num_send_neighbors = d.num_recv_neighbors = 7
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    si, gp = np.ascontiguousarray(send, dtype=np.int32), np.ascontiguousarray(recv, dtype=np.int32)
    d.send_neighbors = d.recv_neighbors = nb.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    d.send_offsets = d.recv_offsets = ip(cuts)
    d.send_indices, d.ghost_positions = ip(si), ip(gp)
    comm = Comm.single()
    h = ctypes.c_void_p()
    _lib.check(L.wf_updater_create(comm._h, ctypes.byref(d), ctypes.byref(h)))
    Kref = oracle.StiffnessOperator(om, p)
    xg = np.random.default_rng(5).uniform(-1, 1, om.ndofs)
    yg = np.zeros(om.ndofs)
    Kref(xg, yg)
    for structured in (True, False):
        part.V.structured = structured
        K = w.StiffnessOperator(part.V, p, {"c0": 1500.0}, tuning={"kernel": "march"})
        assert K.set_ghost_dofs(gp)
        x = torch.from_numpy(np.where(owned, xg[l2g], 0.0)).to(gpu)
        y = torch.zeros_like(x)
        _lib.check(L.wf_op_apply_overlapped(K._h, h, _ptr(x), _ptr(y), _stream(x)))
        torch.cuda.synchronize()
        assert relerr(y.cpu().numpy()[owned], yg[l2g[owned]]) <= TOL, structured
    _lib.check(L.wf_updater_destroy(h))
    comm.close()


def test_cxx_updater_several_segments(gpu, tmp_path):
    """The same through wavehip::VectorUpdater<double> (include/wavehip.hpp): tests/cxx/updater_segments.cpp."""
    exe = str(tmp_path / "updater_segments")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "updater_segments.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "wave_fenics_amd"), "-lwavehip",
                           "-Wl,-rpath," + os.path.join(ROOT, "wave_fenics_amd")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout + r.stderr


def test_file_rendezvous_with_polling_rank(gpu, tmp_path):
    """wf_comm_rendezvous_file: rank 1 polls (a thread here) while rank 0 publishes late; both end
    up with the same 128-byte id; a rank that never finds the file times out with an error, not a
    hang.  Then wf_comm_create_from_file on one rank (publishes, creates, removes the file)."""
    from wave_fenics_amd import _lib
    from wave_fenics_amd.comm import Comm
    L = _lib.lib()
    path = str(tmp_path / "rendezvous_id")
    got = {}

    def poller():
        buf = ctypes.create_string_buffer(_lib.WF_COMM_ID_BYTES)
        t0 = time.time()
        got["rc"] = L.wf_comm_rendezvous_file(path.encode(), 1, 20.0, buf)
        got["dt"] = time.time() - t0
        got["id"] = buf.raw

    th = threading.Thread(target=poller)
    th.start()
    time.sleep(0.4)                      # rank 0 is late
    assert not os.path.exists(path)
    buf0 = ctypes.create_string_buffer(_lib.WF_COMM_ID_BYTES)
    _lib.check(L.wf_comm_rendezvous_file(path.encode(), 0, 20.0, buf0))
    th.join(30.0)
    assert not th.is_alive() and got["rc"] == 0 and got["dt"] >= 0.3
    assert got["id"] == buf0.raw and any(b != 0 for b in buf0.raw)
    os.unlink(path)
    bufx = ctypes.create_string_buffer(_lib.WF_COMM_ID_BYTES)
    assert L.wf_comm_rendezvous_file(str(tmp_path / "never").encode(), 1, 0.2, bufx) == -5     # WF_ERR_COMM
    assert b"timed out" in L.wf_last_error()
    c = Comm.from_file(path, 0, 1)
    assert c.rccl_version() >= 20000 and not os.path.exists(path)
    c.close()


@pytest.mark.parametrize("p", [2, 4])
@pytest.mark.parametrize("kind", ["glued_rotated", "random_orient"])
def test_lattice_renumbering_keeps_the_operator(gpu, oracle, kind, p):
    """wf_lattice_numbering (setup-time renumbering option) on a re-oriented mesh whose dofs were numbered at
    random: the renumbered space is a relabelling -- K x equals the oracle's on the original space, moved by
    the permutation -- and runs the marching kernel."""
    import wave_fenics_amd as w
    from wave_fenics_amd import mesh_io
    mesh, _ = build_mesh(kind, p)
    V = mesh_io.create_functionspace(mesh, p)
    om = oracle_mesh(oracle, mesh, V)
    K = oracle.StiffnessOperator(om, p)
    rng = np.random.default_rng(p)
    x = rng.uniform(-1, 1, V.ndofs)
    yref = np.zeros(V.ndofs)
    K(x, yref)
    scr = rng.permutation(V.ndofs).astype(np.int32)
    Vs = w.renumber(V, scr)
    new = w.lattice_numbering(Vs)
    assert sorted(new.tolist()) == list(range(V.ndofs))
    Vn = w.renumber(Vs, new)
    both = new[scr]                                   # original index -> final index
    xn = np.empty_like(x)
    xn[both] = x
    op = w.StiffnessOperator(Vn, p, {"c0": 1500.0}, structured=False, tuning={"kernel": "march"})
    assert op.kernel == "march_idx"
    y = dev(np.zeros(V.ndofs), gpu)
    op(dev(xn, gpu), y)
    assert relerr(y.cpu().numpy()[both], yref) <= TOL
