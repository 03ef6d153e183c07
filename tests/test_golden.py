"""Frozen self-generated fixtures (tests/golden/, see make_fixtures.py): the live
oracle must reproduce them (CPU), and the HIP kernels must match them (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("name", ["selfgen_hex_p2_n3.npz", "selfgen_hex_p4_n2.npz"])
def test_oracle_reproduces_hex_fixture(oracle, name):
    f = load(name)
    p, n = int(f["p"]), tuple(int(v) for v in f["n"])
    mesh = oracle.create_box(n, p, perturb=0.2, seed=42)
    assert np.array_equal(mesh.x, f["verts"])
    K = oracle.StiffnessOperator(mesh, p)
    assert np.array_equal(K.G, f["G"]) and np.array_equal(K.detJ, f["detJ"])
    y = np.zeros(mesh.ndofs)
    K(f["x"], y)
    assert np.abs(y - f["Kx"]).max() <= 1e-15 * np.abs(f["Kx"]).max()
    m = np.zeros(mesh.ndofs)
    oracle.MassOperatorCPU(mesh, p)(np.ones(mesh.ndofs), m)
    assert np.array_equal(m, f["m"])


def test_oracle_reproduces_rk4_fixture(oracle):
    f = load("selfgen_rk4_p2_n4_5steps.npz")
    p, n = int(f["p"]), int(f["n"])
    mesh = oracle.create_box(n, p, hi=(0.01, 0.01, 0.01))
    eqn = oracle.LinearGLLOpt(mesh, p, 1500.0, 0.5e6, 6e4)
    dt, spp = oracle.cfl_time_step(mesh, p, 1500.0, 0.5e6, CFL=0.25)
    assert dt == float(f["dt"]) and spp == int(f["steps_per_period"])
    eqn.init()
    eqn.rk4(0.0, 5 * dt - 1e-13, dt)
    assert np.abs(eqn.u_n - f["u"]).max() <= 1e-13 * np.abs(f["u"]).max()
    assert np.abs(eqn.v_n - f["v"]).max() <= 1e-13 * np.abs(f["v"]).max()


def test_oracle_reproduces_tet_fixture(oracle):
    from oracle import tet_oracle
    f = load("selfgen_tet_p3.npz")
    p, n = int(f["p"]), tuple(int(v) for v in f["n"])
    mesh = tet_oracle.create_kuhn_box(n, p, perturb=0.2)
    y = np.zeros(mesh.ndofs)
    tet_oracle.TetStiffnessOperator(mesh, p)(f["x"], y)
    assert np.abs(y - f["Kx"]).max() <= 1e-14 * np.abs(f["Kx"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["selfgen_hex_p2_n3.npz", "selfgen_hex_p4_n2.npz"])
def test_hip_matches_hex_fixture(name):
    import torch
    import wave_fenics_amd as w
    f = load(name)
    p, n = int(f["p"]), tuple(int(v) for v in f["n"])
    dev = torch.device("cuda", 0)
    mesh = w.create_box(n, perturb=0.2, seed=42)
    V = w.create_functionspace(mesh, p)
    x = torch.from_numpy(f["x"]).to(dev)
    for kw in (dict(structured=True), dict(structured=False), dict(structured=False, G=f["G"])):
        y = torch.zeros_like(x)
        w.StiffnessOperator(V, p, {"c0": 1500.0}, **kw)(x, y)
        assert np.abs(y.cpu().numpy() - f["Kx"]).max() <= 1e-12 * np.abs(f["Kx"]).max()
    G, detJ = w.precompute_geometric_data(mesh, p)
    assert np.abs(G - f["G"]).max() <= 1e-14 * np.abs(f["G"]).max()
    assert np.abs(detJ - f["detJ"]).max() <= 1e-14 * np.abs(f["detJ"]).max()
    m = torch.zeros_like(x)
    w.MassOperatorLumped(V, p)(torch.ones_like(x), m)
    assert np.abs(m.cpu().numpy() - f["m"]).max() <= 1e-13 * f["m"].max()


@pytest.mark.gpu
def test_hip_matches_tet_fixture():
    import torch
    from wave_fenics_amd import tet
    f = load("selfgen_tet_p3.npz")
    p, n = int(f["p"]), tuple(int(v) for v in f["n"])
    dev = torch.device("cuda", 0)
    V = tet.create_kuhn_box(n, p, perturb=0.2)
    y = torch.zeros(V.ndofs, dtype=torch.float64, device=dev)
    tet.TetStiffnessOperator(V, p)(torch.from_numpy(f["x"]).to(dev), y)
    assert np.abs(y.cpu().numpy() - f["Kx"]).max() <= 1e-12 * np.abs(f["Kx"]).max()
