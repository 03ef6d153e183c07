"""Native RCCL ghost exchange (wf_comm_* / wf_updater_* / wf_op_apply_overlapped,
csrc/comm.hip) on ONE MI355X: a one-rank communicator on a periodic partition is
its own neighbour, so every update is a real grouped ncclSend/ncclRecv to self
through device buffers -- the code path the multi-GPU run uses, with the same
index lists.  The answer is the CPU oracle on the periodic mesh
(oracle.make_periodic).  Replaces demo/gpu_scatter_mpi/VectorUpdater.hpp:106-208.

Tolerances: index work bit-exact; one apply 1e-12; 20 RK4 steps 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import wave_fenics_amd as w
    w.lib()
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def comm(gpu):
    from wave_fenics_amd.comm import Comm
    c = Comm.single()
    assert c.rccl_version() >= 20000
    yield c
    c.close()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def periodic_setup(oracle, n, p, periodic, perturb, hi=(1.0, 1.0, 1.0)):
    from wave_fenics_amd.distributed import create_distributed_box
    part = create_distributed_box(n, p, 1, 0, hi=hi, perturb=perturb, periodic=periodic, build_dofmap=True)
    om = oracle.create_box(n, p, hi=hi, perturb=perturb)
    assert np.array_equal(om.x, part.mesh.x)
    l2g = oracle.make_periodic(om, periodic)          # local lattice index -> periodic dof number
    return part, om, l2g


def test_allreduce_and_barrier(gpu, comm):
    import torch
    t = torch.arange(5, dtype=torch.float64, device=gpu)
    comm.allreduce(t, "sum")
    comm.allreduce(t, "max")
    comm.barrier()
    assert np.array_equal(t.cpu().numpy(), np.arange(5.0))


@pytest.mark.parametrize("periodic", [(True, False, False), (True, True, False), (True, True, True), (False, False, True)])
def test_self_exchange_bit_exact(gpu, comm, oracle, periodic):
    """The reference's own procedure (demo/gpu_scatter_mpi/main.cpp:97: fill, update,
    look at the ghosts): forward update copies owner values into the ghosts, reverse
    update adds ghost values into the owners -- bit for bit."""
    import torch
    from wave_fenics_amd.distributed import VectorUpdater
    part, om, l2g = periodic_setup(oracle, (3, 2, 2), 3, periodic, 0.0)
    vu = VectorUpdater(part, device=gpu, comm=comm)
    assert vu.transport == "native" and vu.active
    owned = part.owned_mask()
    N = l2g.size
    rng = np.random.default_rng(3)
    xg = rng.uniform(-1, 1, om.ndofs)
    xl = np.where(owned, xg[l2g], -7.0)               # ghosts start with a wrong value
    x = torch.from_numpy(xl).to(gpu)
    vu.update_fwd(x)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy(), xg[l2g])
    # reverse: y_owner += sum of the ghost copies
    yl = rng.integers(-8, 8, N).astype(np.float64)    # small integers: sums are exact in any order
    y = torch.from_numpy(yl).to(gpu)
    vu.update_rev(y)
    torch.cuda.synchronize()
    expect = np.zeros(om.ndofs)
    np.add.at(expect, l2g, yl)
    assert np.array_equal(y.cpu().numpy()[owned], expect[l2g[owned]])
    # begin/end split (VectorUpdater.hpp:106-143) with independent work between them
    x2 = torch.from_numpy(xl).to(gpu)
    z = torch.zeros(1 << 20, dtype=torch.float64, device=gpu)
    vu.update_fwd_begin(x2)
    z += 1.0
    vu.update_fwd_end(x2)
    torch.cuda.synchronize()
    assert np.array_equal(x2.cpu().numpy(), xg[l2g]) and float(z.sum()) == float(1 << 20)


@pytest.mark.parametrize("p,n,periodic,perturb", [(2, (4, 3, 3), (True, False, False), 0.2),
                                                  (4, (6, 3, 3), (True, True, False), 0.2),
                                                  (4, (3, 4, 12), (True, True, True), 0.15),
                                                  (3, (3, 3, 4), (False, True, True), 0.2),
                                                  (6, (5, 2, 4), (True, True, True), 0.15)])
def test_periodic_stiffness_vs_oracle(gpu, comm, oracle, p, n, periodic, perturb):
    """scatter_fwd(x); y += K x; scatter_rev(y) (LinearGLL.hpp:164-176) against the
    oracle's stiffness operator on the periodic mesh: arbitrary-dofmap kernel and box kernel
    (unsplit sequence), and wf_op_apply_overlapped (interior / interface split, both halo
    directions hidden) for BOTH -- the box operator split by ghost faces, the arbitrary-dofmap
    operator by ghost dofs (VectorUpdater.hpp:106-143,157-199 works on any IndexMap)."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd.distributed import VectorUpdater, overlapped_apply
    part, om, l2g = periodic_setup(oracle, n, p, periodic, perturb)
    vu = VectorUpdater(part, device=gpu, comm=comm)
    owned = part.owned_mask()
    Kref = oracle.StiffnessOperator(om, p)
    xg = np.random.default_rng(11).uniform(-1, 1, om.ndofs)
    yg = np.zeros(om.ndofs)
    Kref(xg, yg)
    xl = np.where(owned, xg[l2g], 0.0)
    for mode in ("generic", "box", "overlapped", "overlapped_generic"):
        part.V.structured = mode in ("box", "overlapped")
        K = w.StiffnessOperator(part.V, p, {"c0": 1500.0}, tuning={"kernel": "march"})
        assert K.kernel == ("march_box" if part.V.structured else "march_idx")
        x = torch.from_numpy(xl).to(gpu)
        y = torch.zeros_like(x)
        if mode == "overlapped":
            assert K.set_ghost_faces(*[bool(v) for v in part.owned_lo])
            overlapped_apply(K, vu, x, y)
        elif mode == "overlapped_generic":
            assert K.set_ghost_dofs(vu.h_ghost_pos)
            assert K.info.items_interface > 0
            overlapped_apply(K, vu, x, y)
        else:
            vu.update_fwd(x)
            K(x, y)
            vu.update_rev(y)
        torch.cuda.synchronize()
        err = relerr(y.cpu().numpy()[owned], yg[l2g[owned]])
        assert err <= 1e-12, (mode, err)


@pytest.mark.parametrize("lz,lz0", [(7, 2), (5, 1), (6, 6)])
def test_overlapped_apply_short_first_segment(gpu, comm, oracle, lz, lz0):
    """The split operator's z segmentation [0, lz0), then pieces of lz layers (the short first
    segment keeps the work that waits for the z halo small): wf_op_apply_overlapped on a fully
    periodic mesh with the segment lengths forced, against the oracle; lz0 = lz is the unsplit layout."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd.distributed import VectorUpdater, overlapped_apply
    p, n = 4, (3, 4, 13)
    part, om, l2g = periodic_setup(oracle, n, p, (True, True, True), 0.15)
    vu = VectorUpdater(part, device=gpu, comm=comm)
    owned = part.owned_mask()
    Kref = oracle.StiffnessOperator(om, p)
    xg = np.random.default_rng(12).uniform(-1, 1, om.ndofs)
    yg = np.zeros(om.ndofs)
    Kref(xg, yg)
    part.V.structured = True
    K = w.StiffnessOperator(part.V, p, {"c0": 1500.0}, tuning={"lz": lz, "lz0": lz0})
    x = torch.from_numpy(np.where(owned, xg[l2g], 0.0)).to(gpu)
    y = torch.zeros_like(x)
    assert K.set_ghost_faces(*[bool(v) for v in part.owned_lo])
    overlapped_apply(K, vu, x, y)
    torch.cuda.synchronize()
    assert relerr(y.cpu().numpy()[owned], yg[l2g[owned]]) <= 1e-12
    # the parts together are the whole operator, whatever the segmentation
    y2 = torch.zeros_like(x)
    K(x, y2)      # unsplit apply (uniform segments) on the halo-updated x
    vu.update_rev(y2)
    torch.cuda.synchronize()
    assert relerr(y2.cpu().numpy()[owned], yg[l2g[owned]]) <= 1e-12


def test_periodic_rk4_vs_oracle(gpu, comm, oracle):
    """Full RK4 loop with the native exchange every stage (cfg4's code path on one
    GPU): source on x = lo, absorbing x = hi, periodic in y and z."""
    from wave_fenics_amd.distributed import VectorUpdater, boundary_tags
    from wave_fenics_amd.linear_gll import LinearGLLOpt
    p, n, periodic, hi = 4, (4, 3, 6), (False, True, True), (0.01, 0.01, 0.01)
    part, om, l2g = periodic_setup(oracle, n, p, periodic, 0.0, hi=hi)
    vu = VectorUpdater(part, device=gpu, comm=comm)
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    dt, _ = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.25)
    ref.init()
    ref.rk4(0.0, 20 * dt - 1e-13, dt)
    for fused in (False, True):
        eqn = LinearGLLOpt(part.V, p, 1500.0, 0.5e6, 6e4, updater=vu, tags=boundary_tags(part), device=gpu)
        assert eqn._split
        eqn.init()
        (eqn.rk4_fused if fused else eqn.rk4)(0.0, 20 * dt - 1e-13, dt)
        assert relerr(eqn.u_n.cpu().numpy(), ref.u_n[l2g]) <= 1e-9     # ghosts included (final scatter_fwd)
        assert relerr(eqn.v_n.cpu().numpy(), ref.v_n[l2g]) <= 1e-9
    # boundary= handed over explicitly as RANK-LOCAL sets (ghost boundary dofs included, rank-local facet
    # masses): they go through the same accumulate-to-owner step as the tag-derived ones, and
    # an arbitrary-dofmap space is split by ghost dofs
    from wave_fenics_amd.linear_gll import facet_lumped_mass
    tags = boundary_tags(part)
    sets = (facet_lumped_mass(part.V, tags, 1), facet_lumped_mass(part.V, tags, 2))
    assert (~part.owned_mask()[sets[1][0]]).any(), "the test needs ghost dofs in the rank-local boundary set"
    part.V.structured = False
    eqn = LinearGLLOpt(part.V, p, 1500.0, 0.5e6, 6e4, updater=vu, boundary=sets, device=gpu)
    assert eqn._split and eqn.stiff_op.kernel == "march_idx"
    eqn.init()
    eqn.rk4_fused(0.0, 20 * dt - 1e-13, dt)
    assert relerr(eqn.u_n.cpu().numpy(), ref.u_n[l2g]) <= 1e-9
    assert relerr(eqn.v_n.cpu().numpy(), ref.v_n[l2g]) <= 1e-9
    part.V.structured = True


def test_cxx_updater_and_multi_rank_driver(gpu, oracle, tmp_path):
    """The C++ host side over the same C ABI: wavehip::Comm / VectorUpdater<double>
    (examples/scatter_demo.cpp = demo/gpu_scatter_mpi/main.cpp) and the partition-aware
    LinearGLLOpt (examples/planar3d.cpp) on a periodic one-rank partition, i.e. with a
    real RCCL exchange every stage; fused and reference-order RK4 against the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "bin")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), f"OUT={out}"])
    env = dict(os.environ, WF_COMM_FILE=str(tmp_path / "comm_id"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([os.path.join(out, "scatter_demo"), "--size", "6", "--degree", "3", "--reps", "5", "--periodic", "xyz"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout + r.stderr
    p, N, L = 2, 6, 0.01
    from wave_fenics_amd.distributed import create_distributed_box
    part = create_distributed_box(N, p, 1, 0, hi=(L, L, L), periodic=(False, True, True))
    om = oracle.create_box(N, p, hi=(L, L, L))
    l2g = oracle.make_periodic(om, (False, True, True))
    ref = oracle.LinearGLLOpt(om, p, 1500.0, 0.5e6, 6e4)
    dt, spp = oracle.cfl_time_step(om, p, 1500.0, 0.5e6, CFL=0.25)
    ref.init()
    ref.rk4(0.0, 30 * dt - 1e-13, dt)
    for extra in ([], ["--reference-order"]):
        dump = str(tmp_path / "uv.bin")
        r = subprocess.run([os.path.join(out, "planar3d"), "--size", str(N), "--degree", str(p), "--cfl", "0.25", "--steps", "30",
                            "--length", str(L), "--periodic", "yz", "--dump", dump] + extra,
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Steps taken: 30" in r.stdout and f"Degrees of freedom: {om.ndofs}" in r.stdout
        uv = np.fromfile(dump, dtype=np.float64)
        n = part.V.ndofs
        assert uv.size == 2 * n
        assert relerr(uv[:n], ref.u_n[l2g]) <= 1e-9
        assert relerr(uv[n:], ref.v_n[l2g]) <= 1e-9
