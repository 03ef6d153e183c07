#!/usr/bin/env python3
"""Generate the SELF-GENERATED regression fixtures in this directory.

These are NOT reference-generated golden vectors: the reference cannot be built
or run offline and ships no fixtures (parity unpinned).  They
are outputs of this repo's own CPU restatement (oracle/), committed so that (a) an
accidental change of the oracle shows up as a diff against a frozen answer and
(b) the GPU kernels are checked against data files as well as against the live
oracle.  Re-run only when the oracle is changed on purpose:

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tet_oracle, wave_oracle as o  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hex_case(p, n, seed):
    mesh = o.create_box(n, p, perturb=0.2, seed=42)
    K = o.StiffnessOperator(mesh, p)
    M = o.MassOperatorCPU(mesh, p)
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, mesh.ndofs)
    Kx = np.zeros(mesh.ndofs)
    K(x, Kx)
    m = np.zeros(mesh.ndofs)
    M(np.ones(mesh.ndofs), m)
    return dict(p=p, n=np.array(n), x=x, Kx=Kx, m=m, G=K.G, detJ=K.detJ, verts=mesh.x)


def rk4_case():
    p, n = 2, 4
    mesh = o.create_box(n, p, hi=(0.01, 0.01, 0.01))
    eqn = o.LinearGLLOpt(mesh, p, 1500.0, 0.5e6, 6e4)
    dt, spp = o.cfl_time_step(mesh, p, 1500.0, 0.5e6, CFL=0.25)
    eqn.init()
    eqn.rk4(0.0, 5 * dt - 1e-13, dt)
    return dict(p=p, n=n, dt=dt, steps_per_period=spp, u=eqn.u_n.copy(), v=eqn.v_n.copy())


def tet_case():
    p, n = 3, (2, 2, 1)
    mesh = tet_oracle.create_kuhn_box(n, p, perturb=0.2)
    K = tet_oracle.TetStiffnessOperator(mesh, p)
    x = np.random.default_rng(9).uniform(-1, 1, mesh.ndofs)
    Kx = np.zeros(mesh.ndofs)
    K(x, Kx)
    return dict(p=p, n=np.array(n), x=x, Kx=Kx)


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "selfgen_hex_p2_n3.npz"), **hex_case(2, (3, 3, 3), 1))
    np.savez_compressed(os.path.join(HERE, "selfgen_hex_p4_n2.npz"), **hex_case(4, (2, 2, 2), 2))
    np.savez_compressed(os.path.join(HERE, "selfgen_rk4_p2_n4_5steps.npz"), **rk4_case())
    np.savez_compressed(os.path.join(HERE, "selfgen_tet_p3.npz"), **tet_case())
    print("written")
