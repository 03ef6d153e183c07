"""Known-answer tests for the tetrahedral restatement (oracle/tet_oracle.py)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def tet(oracle):
    from oracle import tet_oracle
    return tet_oracle


def test_tet_quadrature(tet):
    for m in (1, 2, 3, 4):
        X, W = tet.tet_quadrature(m)
        assert abs(W.sum() - 1.0 / 6.0) < 1e-15
        assert (X >= 0).all() and (X.sum(axis=1) <= 1 + 1e-15).all()
        # exact for monomials up to degree 2m-1:  int x^a y^b z^c = a! b! c! / (a+b+c+3)!
        from math import factorial as f
        for a in range(2 * m):
            for b in range(2 * m - a):
                c = 2 * m - 1 - a - b
                exact = f(a) * f(b) * f(c) / f(a + b + c + 3)
                assert abs(np.dot(W, X[:, 0] ** a * X[:, 1] ** b * X[:, 2] ** c) - exact) < 1e-15


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_tet_basis(tet, p):
    nodes = tet.tet_nodes(p)
    assert nodes.shape[0] == (p + 1) * (p + 2) * (p + 3) // 6
    phi, dphi = tet.tabulate_tet(p, nodes / float(p))
    assert np.abs(phi - np.eye(nodes.shape[0])).max() < 1e-10        # nodal
    X, W = tet.tet_quadrature(p)
    phi, dphi = tet.tabulate_tet(p, X)
    assert np.abs(phi.sum(axis=1) - 1).max() < 1e-11                 # partition of unity
    assert np.abs(dphi.sum(axis=2)).max() < 1e-9


@pytest.mark.parametrize("p,n,perturb", [(1, 2, 0.2), (2, 2, 0.2), (4, (2, 1, 1), 0.2), (3, 2, 0.0)])
def test_tet_stiffness_kats(tet, p, n, perturb):
    mesh = tet.create_kuhn_box(n, p, perturb=perturb)
    assert mesh.ncells == 6 * np.prod(mesh.n)
    # conforming: the dofs of all cells cover the lattice exactly
    assert np.array_equal(np.unique(mesh.dofmap), np.arange(mesh.ndofs))
    K = tet.TetStiffnessOperator(mesh, p)
    assert abs(K.detJ.sum() - 1.0) < 1e-13                            # volumes add up
    c02 = 1500.0 ** 2
    N = mesh.ndofs
    y = np.zeros(N)
    K(np.ones(N), y)
    assert np.abs(y).max() < 1e-8 * c02
    Xd = tet.dof_coordinates(mesh)
    y[:] = 0
    K(Xd[:, 0].copy(), y)
    assert abs(np.dot(Xd[:, 0], y) / (-c02) - 1.0) < 1e-9
    assert abs(np.dot(Xd[:, 1], y) / c02) < 1e-7
    rng = np.random.default_rng(0)
    u, v = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
    Ku, Kv = np.zeros(N), np.zeros(N)
    K(u, Ku)
    K(v, Kv)
    assert abs(np.dot(v, Ku) - np.dot(u, Kv)) < 1e-11 * abs(np.dot(v, Ku))


def test_tet_matches_hex_on_linears(tet, oracle):
    """Both discretisations reproduce the exact energy of a quadratic field on an
    affine mesh: u = x^2 + y z  ->  int |grad u|^2 = 4/3 + 1/3 + 1/3 = 2 (P2 exact)."""
    p = 2
    tm = tet.create_kuhn_box(2, p)
    hm = oracle.create_box(2, p)
    for mesh, K, X in ((tm, tet.TetStiffnessOperator(tm, p), tet.dof_coordinates(tm)),
                       (hm, oracle.StiffnessOperator(hm, p), oracle.dof_coordinates(hm))):
        u = X[:, 0] ** 2 + X[:, 1] * X[:, 2]
        y = np.zeros(mesh.ndofs)
        K(u, y)
        assert abs(np.dot(u, y) / (-1500.0 ** 2) - 2.0) < 1e-9
