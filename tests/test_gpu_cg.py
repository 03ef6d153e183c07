"""Matrix-free CG (wf_cg; BP1 of demo/gpu_cg, CUDA/cg.hpp:38-121) on the MI355X:
against a dense numpy solve of the oracle's mass matrix, against a numpy CG for the
iteration count, through the Python callback, on a periodic partition with the
native RCCL updater (halo + all-reduce inside the solver), and residual /
iteration-count known answers at larger size."""
import os

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


def numpy_cg(A, b, kmax, rtol):
    """Textbook CG with the reference's stopping rule (cg.hpp:103)."""
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rr0 = rr = r @ r
    k = 0
    while k < kmax:
        k += 1
        y = A @ p
        alpha = rr / (p @ y)
        x += alpha * p
        r -= alpha * y
        rr_new = r @ r
        if rr_new / rr0 < rtol * rtol:
            break
        p = r + (rr_new / rr) * p
        rr = rr_new
    return x, k


def assemble(oracle, om, phi, detJ):
    """The oracle's mass matrix, column by column (A e_j)."""
    A = np.zeros((om.ndofs, om.ndofs))
    e, col = np.zeros(om.ndofs), np.zeros(om.ndofs)
    for j in range(om.ndofs):
        e[:] = 0.0
        e[j] = 1.0
        col[:] = 0.0
        oracle.dense_mass_apply(om, phi, detJ, e, col)
        A[:, j] = col
    return A


def dense_mass_setup(oracle, p, n, perturb=0.2):
    """Consistent (non-diagonal) mass: GLL-warped basis, Gauss rule of degree 2P."""
    import wave_fenics_amd as w
    om = oracle.create_box(n, p, perturb=perturb)
    mesh = w.create_box(n, perturb=perturb)
    V = w.create_functionspace(mesh, p)
    pts, wts, phi1, phi, X, W = oracle.tabulate_mass_tables(p, "gll", "gauss_jacobi", 2 * p)
    detJ = oracle.compute_detJ_generic(om, X, W)
    return om, V, phi1, np.abs(detJ), assemble(oracle, om, phi, np.abs(detJ))


@pytest.mark.parametrize("p,n", [(2, (3, 3, 2)), (3, (2, 2, 2))])
def test_cg_dense_mass_vs_numpy(gpu, oracle, p, n):
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd import la
    om, V, phi1, detJ, A = dense_mass_setup(oracle, p, n)
    assert np.abs(A - A.T).max() <= 1e-14 * np.abs(A).max()
    op = w.MassOperator(V, p, phi1, detJ)
    rng = np.random.default_rng(8)
    b = rng.uniform(-1, 1, om.ndofs)
    xs = np.linalg.solve(A, b)
    x_np, k_np = numpy_cg(A, b, 200, 1e-10)
    for mode in ("handle", "callback"):
        x = torch.zeros(om.ndofs, dtype=torch.float64, device=gpu)
        Aop = op if mode == "handle" else (lambda v, y: op(v, y))
        its, res = la.cg(x, torch.from_numpy(b).to(gpu), Aop, kmax=200, rtol=1e-10)
        # the operator sums its contributions with atomics (order varies from run to run): near convergence the
        # iteration count of a 126-iteration solve moves by a few steps with the rounding
        assert abs(its - k_np) <= max(3, k_np // 25), (its, k_np)
        assert res < 1e-10
        assert np.abs(x.cpu().numpy() - xs).max() <= 1e-7 * np.abs(xs).max()
    # non-zero initial guess: the iteration starts from r = b - A x0
    x = torch.from_numpy(0.5 * xs + 0.01 * np.abs(xs).max()).to(gpu)
    its, res = la.cg(x, torch.from_numpy(b).to(gpu), op, kmax=200, rtol=1e-10)
    assert res < 1e-10 and np.abs(x.cpu().numpy() - xs).max() <= 1e-7 * np.abs(xs).max()
    # kmax is honoured (reference default 50 iterations, cg.hpp:42)
    x = torch.zeros(om.ndofs, dtype=torch.float64, device=gpu)
    its, res = la.cg(x, torch.from_numpy(b).to(gpu), op, kmax=3, rtol=1e-14)
    assert its == 3 and res > 1e-14


def test_cg_periodic_partition_native_rccl(gpu, oracle):
    """CG with the halo exchange and the scalar all-reduces inside the solver: one rank,
    periodic in x and z, the RCCL updater exchanging with itself.  The solution of the
    periodic problem (oracle mass matrix on the periodic mesh) is recovered on the owned
    entries."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd import la
    from wave_fenics_amd.comm import Comm
    from wave_fenics_amd.distributed import VectorUpdater, create_distributed_box
    p, n, per = 2, (3, 2, 3), (True, False, True)
    part = create_distributed_box(n, p, 1, 0, perturb=0.0, periodic=per, build_dofmap=True)
    om = oracle.create_box(n, p)
    l2g = oracle.make_periodic(om, per)
    pts, wts, phi1, phi, X, W = oracle.tabulate_mass_tables(p, "gll", "gauss_jacobi", 2 * p)
    detJ = np.abs(oracle.compute_detJ_generic(om, X, W))
    A = assemble(oracle, om, phi, detJ)
    comm = Comm.single()
    vu = VectorUpdater(part, device=gpu, comm=comm)
    part.V.structured = False
    op = w.MassOperator(part.V, p, phi1, detJ)       # local (non-periodic) operator; periodicity comes from the exchange
    owned = part.owned_mask()
    bg = np.random.default_rng(2).uniform(-1, 1, om.ndofs)
    xs = np.linalg.solve(A, bg)
    b = torch.from_numpy(np.where(owned, bg[l2g], 0.0)).to(gpu)
    x = torch.zeros_like(b)
    its, res = la.cg(x, b, op, kmax=300, rtol=1e-9, updater=vu)
    _, k_np = numpy_cg(A, bg, 300, 1e-9)
    assert abs(its - k_np) <= max(3, k_np // 25) and res < 1e-9, (its, k_np, res)   # (atomics: see test_cg_dense_mass_vs_numpy)
    assert np.abs(x.cpu().numpy()[owned] - xs[l2g[owned]]).max() <= 1e-7 * np.abs(xs).max()
    comm.close()


def test_cg_lumped_mass_cfg_size(gpu):
    """Known answers at BASELINE cfg2 size (P4, 54^3 cells, 10.2 M dofs): the lumped mass
    of the uniform box is diagonal with few distinct entries (products of summed GLL
    weights), so CG reaches x = b / m in about that many iterations."""
    import torch
    import wave_fenics_amd as w
    from wave_fenics_amd import la
    p, N = 4, 54
    mesh = w.create_box(N)
    V = w.create_functionspace(mesh, p, build_dofmap=False)
    M = w.MassOperatorLumped(V, p)
    n = V.ndofs
    m = torch.zeros(n, dtype=torch.float64, device=gpu)
    M(torch.ones(n, dtype=torch.float64, device=gpu), m)
    ndistinct = int(torch.unique(torch.round(m / m.max() * 1e12)).numel())
    assert ndistinct <= 64
    b = torch.sin(torch.arange(n, dtype=torch.float64, device=gpu) * 0.001) + 1.5
    x = torch.zeros_like(b)
    its, res = la.cg(x, b, M, kmax=100, rtol=1e-10)
    assert its <= 2 * ndistinct and res < 1e-10, (its, ndistinct)     # exact arithmetic: its <= ndistinct
    assert float(((x - b / m).abs() / (b / m).abs()).max()) <= 1e-8


@pytest.mark.parametrize("opname,degree", [("lumped", 2), ("dense", 2), ("dense", 4)])
def test_cxx_cg_demo(gpu, tmp_path, opname, degree):
    """examples/cg_demo.cpp = demo/gpu_cg/main.cpp (BP1: mass-operator CG, f = x0 + 4, kmax 50,
    rtol 1e-4): the C++ host side converges within kmax and reproduces f."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "bin")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), f"OUT={out}", "CXXFLAGS=-O1 -std=c++17"])
    # dense: the consistent mass matrix has condition number ~ kappa_1d^3 (thousands), so the demo's
    # rtol 1e-4 leaves a visible error; tighten it to check the solution, keep the defaults for lumped
    extra = ["--rtol", "1e-10", "--kmax", "400"] if opname == "dense" else []
    r = subprocess.run([os.path.join(out, "cg_demo"), "--size", "8", "--degree", str(degree), "--op", opname] + extra,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    its = int(re.search(r"its = (\d+)", r.stdout).group(1))
    err = float(re.search(r"max \|u - f\|: (\S+)", r.stdout).group(1))
    assert 1 <= its <= (400 if opname == "dense" else 50)
    assert err < 1e-6, r.stdout
