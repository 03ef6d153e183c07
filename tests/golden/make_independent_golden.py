#!/usr/bin/env python3
"""Independent derivation of the quantities the oracle restates, for pinning it.

The reference ships no fixtures and cannot be built here, so the
oracle stays "parity unpinned" by rule.  What this script adds is a SECOND
derivation that shares no code with oracle/ or with the product: everything is
obtained from the mathematical definitions with sympy (symbolic polynomials,
symbolic differentiation, symbolic 3x3 inverse) and evaluated with mpmath at 40
digits, then rounded once to float64:

  * GLL rule on [0,1], n = 2..8 points: nodes = roots of (1-x^2) P'_{n-1}(x),
    weights = 2 / (n(n-1) P_{n-1}(x)^2); collocation derivative matrix
    D[q][a] = l_a'(x_q) from symbolically differentiated Lagrange polynomials.
    (Basix' GLL rule / GLL-warped Lagrange element, common/operators.hpp:16-24.)
  * On a small box mesh of trilinear hexahedra whose vertices are displaced by a
    closed-form field (boundary vertices slide inside their boundary planes, so
    facets are non-rectangular):
      K x   with K_ij = -c0^2 sum_q grad(phi_i)^T [ w_q |det J| J^-1 J^-T ] grad(phi_j)
            (common/operators.hpp:113-133 + common/precomputation.hpp:83-100),
      m = M 1 with the collocated mass m_i = sum_q w_q |det J|  (operators.hpp:36-40),
      facet masses m_Gamma[i] = sum_facets w_q |dx/ds x dx/dt| for tag 1 (x = 0) and
            tag 2 (every other face) -- the diagonal GLL form of
            demo/cpu_planar3d/forms.ufl:19-24.
    phi_i are tensor products of the symbolic 1-D Lagrange polynomials; J comes from
    symbolic differentiation of the trilinear map.
The -1/0/1 clamps of the reference are inactive on this mesh (asserted below).

Output: tests/golden/independent.json (floats as repr strings, exact round trip).
Run: python tests/golden/make_independent_golden.py      (about a minute)"""
import json
import os

import mpmath as mp
import numpy as np
import sympy as sp

mp.mp.dps = 40
HERE = os.path.dirname(os.path.abspath(__file__))
X = sp.Symbol("x")


def gll_rule(n):
    """n-point Gauss-Lobatto-Legendre rule on [0,1]: (nodes, weights) as mp floats."""
    N = n - 1
    P = sp.legendre(N, X)
    if n == 2:
        interior = []
    else:
        poly = sp.Poly(sp.diff(P, X), X)
        interior = sorted(mp.mpf(sp.N(r, 45)) for r in sp.real_roots(poly))
    nodes = [mp.mpf(-1)] + interior + [mp.mpf(1)]
    Pf = sp.lambdify(X, P, "mpmath")
    w = [2 / (N * (N + 1) * Pf(x) ** 2) for x in nodes]
    return [(x + 1) / 2 for x in nodes], [v / 2 for v in w]


def gauss_rule(m):
    """m-point Gauss-Legendre rule on [0,1]: nodes = roots of P_m, w = 2 / ((1-x^2) P_m'(x)^2)."""
    P = sp.legendre(m, X)
    nodes = sorted(mp.mpf(sp.N(r, 45)) for r in sp.real_roots(sp.Poly(P, X)))
    dP = sp.lambdify(X, sp.diff(P, X), "mpmath")
    w = [2 / ((1 - x * x) * dP(x) ** 2) for x in nodes]
    return [(x + 1) / 2 for x in nodes], [v / 2 for v in w]


def lagrange_table(nodes, pts, derivative):
    """table[q][a] = l_a(pts[q]) (or its derivative), symbolic Lagrange polynomials through `nodes`."""
    n = len(nodes)
    sn = [sp.Float(str(v), 45) for v in nodes]
    T = [[None] * n for _ in pts]
    for a in range(n):
        la = sp.Integer(1)
        for b in range(n):
            if b != a:
                la = la * (X - sn[b]) / (sn[a] - sn[b])
        e = sp.expand(la)
        f = sp.lambdify(X, sp.diff(e, X) if derivative else e, "mpmath")
        for q, x in enumerate(pts):
            T[q][a] = f(x)
    return T


def lagrange_derivative_matrix(nodes):
    """D[q][a] = l_a'(nodes[q]) by symbolic differentiation of the Lagrange polynomials."""
    n = len(nodes)
    sn = [sp.Float(str(v), 45) for v in nodes]
    D = [[None] * n for _ in range(n)]
    for a in range(n):
        la = sp.Integer(1)
        for b in range(n):
            if b != a:
                la = la * (X - sn[b]) / (sn[a] - sn[b])
        dla = sp.lambdify(X, sp.diff(sp.expand(la), X), "mpmath")
        for q in range(n):
            D[q][a] = dla(nodes[q])
    return D


def displaced_vertices(n):
    """Vertices of the n[0] x n[1] x n[2] box on the unit cube, displaced by a closed-form
    field; the component normal to a boundary plane is zero on that plane."""
    nx, ny, nz = n
    out = []
    for c in range(nz + 1):
        for b in range(ny + 1):
            for a in range(nx + 1):
                x, y, z = mp.mpf(a) / nx, mp.mpf(b) / ny, mp.mpf(c) / nz
                s = mp.mpf("0.11")
                dx = s / nx * mp.sin(3 * x + 2 * y + z + 1) * (0 if a in (0, nx) else 1)
                dy = s / ny * mp.cos(x - 2 * y + 3 * z + 2) * (0 if b in (0, ny) else 1)
                dz = s / nz * mp.sin(2 * x + y - 2 * z + 3) * (0 if c in (0, nz) else 1)
                out.append((x + dx, y + dy, z + dz))
    return out


def to_f(a):
    return np.array([[float(v) for v in row] for row in a]) if isinstance(a[0], (list, tuple)) else np.array([float(v) for v in a])


def operators_on_mesh(p, n, c0):
    nodes, wts = gll_rule(p + 1)
    D = to_f(lagrange_derivative_matrix(nodes))
    w1 = to_f(wts)
    x1 = to_f(nodes)
    nn = p + 1
    verts_mp = displaced_vertices(n)
    verts = to_f(verts_mp)
    nx, ny, nz = n
    NX, NY, NZ = p * nx + 1, p * ny + 1, p * nz + 1
    ndofs = NX * NY * NZ
    # symbolic trilinear map and its Jacobian
    xi = sp.symbols("xi0:3")
    Xv = sp.symbols("X0:24")          # vertex v = a + 2b + 4c, component d: X[3v+d]
    xmap = [0, 0, 0]
    for v in range(8):
        bits = (v & 1, (v >> 1) & 1, (v >> 2) & 1)
        N = sp.Integer(1)
        for d in range(3):
            N = N * (xi[d] if bits[d] else 1 - xi[d])
        for d in range(3):
            xmap[d] = xmap[d] + Xv[3 * v + d] * N
    J = sp.Matrix(3, 3, lambda i, j: sp.diff(xmap[i], xi[j]))
    detJ = J.det()
    adj = J.adjugate()                 # J^-1 = adj / det
    Gs = (adj * adj.T)                 # (J^-1 J^-T) det^2
    f_det = sp.lambdify(list(xi) + list(Xv), detJ, "numpy")
    f_G = sp.lambdify(list(xi) + list(Xv), Gs, "numpy")
    f_J = sp.lambdify(list(xi) + list(Xv), J, "numpy")
    # gradient table of the tensor Lagrange basis at the collocated points:
    # d phi_(a,b,c) / d xi_0 at point (i,j,k) = D[i][a] delta_jb delta_kc, etc.  Built from the
    # symbolic derivative matrix, not from any oracle routine.
    I = np.eye(nn)
    dphi = np.zeros((3, nn, nn, nn, nn, nn, nn))       # [d][k][j][i][c][b][a]
    dphi[0] = np.einsum("ia,jb,kc->kjicba", D, I, I)
    dphi[1] = np.einsum("ia,jb,kc->kjicba", I, D, I)
    dphi[2] = np.einsum("ia,jb,kc->kjicba", I, I, D)
    nd = nn ** 3
    dphi = dphi.reshape(3, nd, nd)                      # [d][q][i], q = i + n(j + n k)
    kk, jj, ii = np.meshgrid(np.arange(nn), np.arange(nn), np.arange(nn), indexing="ij")
    qx, qy, qz = x1[ii.reshape(-1)], x1[jj.reshape(-1)], x1[kk.reshape(-1)]
    wq = (w1[ii] * w1[jj] * w1[kk]).reshape(-1)
    rng_x = np.array([np.sin(0.37 * g + 0.11) + 0.25 * np.cos(1.3 * g) for g in range(ndofs)])
    Kx = np.zeros(ndofs)
    m = np.zeros(ndofs)
    mG = {1: np.zeros(ndofs), 2: np.zeros(ndofs)}
    clamp_hits = 0
    for cz in range(nz):
        for cy in range(ny):
            for cx in range(nx):
                vid = [(cx + (v & 1)) + (nx + 1) * ((cy + ((v >> 1) & 1)) + (ny + 1) * (cz + ((v >> 2) & 1))) for v in range(8)]
                Xc = verts[vid].reshape(-1)
                dof = ((p * cx + ii) + NX * ((p * cy + jj) + NY * (p * cz + kk))).reshape(-1)
                det = np.array([f_det(qx[q], qy[q], qz[q], *Xc) for q in range(nd)])
                Gq = np.array([np.array(f_G(qx[q], qy[q], qz[q], *Xc), dtype=float) for q in range(nd)])
                G = Gq * (wq * np.abs(det) / det ** 2)[:, None, None]      # w |det| J^-1 J^-T
                # the reference's clamp (isclose to -1/0/1, rtol 1e-5, atol 1e-8) must be a no-op here:
                # entries inside a window must already equal the clamp value up to rounding noise
                a = np.abs(G)
                clamp_hits += int(((a > 1e-13) & (a <= 1.1e-8)).sum() + ((np.abs(a - 1) > 1e-13) & (np.abs(a - 1) <= 1.2e-5)).sum())
                xe = rng_x[dof]
                grad = np.einsum("dqi,i->qd", dphi, xe)
                flux = -c0 ** 2 * np.einsum("qde,qe->qd", G, grad)
                np.add.at(Kx, dof, np.einsum("dqi,qd->i", dphi, flux))
                np.add.at(m, dof, wq * np.abs(det))
                # exterior facets: surface element from the symbolic Jacobian columns
                cc, ncell = (cx, cy, cz), (nx, ny, nz)
                for axis in range(3):
                    for side in (0, 1):
                        if cc[axis] != (0 if side == 0 else ncell[axis] - 1):
                            continue
                        tag = 1 if (axis == 0 and side == 0) else 2
                        ta, tb = [d for d in range(3) if d != axis]
                        for bq in range(nn):
                            for aq in range(nn):
                                ref = [0.0, 0.0, 0.0]
                                ref[axis] = float(side)
                                ref[ta], ref[tb] = x1[aq], x1[bq]
                                Jm = np.array(f_J(*ref, *Xc), dtype=float)
                                ds = np.linalg.norm(np.cross(Jm[:, ta], Jm[:, tb]))
                                loc = [0, 0, 0]
                                loc[axis] = side * p
                                loc[ta], loc[tb] = aq, bq
                                g = (p * cx + loc[0]) + NX * ((p * cy + loc[1]) + NY * (p * cz + loc[2]))
                                mG[tag][g] += w1[aq] * w1[bq] * ds
    assert clamp_hits == 0, "a G entry falls inside the reference's clamp window; pick another displacement"
    return {"p": p, "n": list(n), "c0": c0, "verts": verts, "x": rng_x, "Kx": Kx, "m": m, "mG1": mG[1], "mG2": mG[2]}


def enc(a):
    a = np.asarray(a, dtype=np.float64)
    return {"shape": list(a.shape), "data": [repr(float(v)) for v in a.reshape(-1)]}


def main():
    out = {"comment": "independent sympy/mpmath derivation; generator: tests/golden/make_independent_golden.py",
           "gll": {}, "mesh_cases": []}
    for n in range(2, 9):
        nodes, wts = gll_rule(n)
        D = lagrange_derivative_matrix(nodes)
        out["gll"][str(n - 1)] = {"points": enc(to_f(nodes)), "weights": enc(to_f(wts)), "D": enc(to_f(D))}
        print("gll", n, flush=True)
    # Gauss-Legendre rules (basix gauss_jacobi on the interval, precompute.hpp:183-184) and the
    # tables of tabulate_1d (precompute.hpp:179-189) / demo/gpu_operator (equispaced basis, degree 2P rule)
    out["gauss"] = {}
    for m in range(1, 9):
        nodes, wts = gauss_rule(m)
        out["gauss"][str(m)] = {"points": enc(to_f(nodes)), "weights": enc(to_f(wts))}
    out["tabulate_1d"] = []
    for p in (2, 3, 4, 6):
        gpts, _ = gauss_rule(p + 1)               # degree 2P -> (2P+2)/2 = P+1 points
        for variant in ("gll_warped", "equispaced"):
            nodes = gll_rule(p + 1)[0] if variant == "gll_warped" else [mp.mpf(a) / p for a in range(p + 1)]
            out["tabulate_1d"].append({"p": p, "variant": variant, "points": enc(to_f(gpts)),
                                       "phi": enc(to_f(lagrange_table(nodes, gpts, 0))),
                                       "dphi": enc(to_f(lagrange_table(nodes, gpts, 1)))})
        print("tabulate_1d", p, flush=True)
    for p, n in [(1, (2, 2, 2)), (2, (2, 2, 2)), (3, (2, 1, 2)), (4, (1, 2, 1))]:
        r = operators_on_mesh(p, n, 1500.0)
        out["mesh_cases"].append({k: (enc(v) if isinstance(v, np.ndarray) else v) for k, v in r.items()})
        print("mesh case", p, n, flush=True)
    with open(os.path.join(HERE, "independent.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
