"""LinearGLLOpt -- the explicit RK4 wave model of common/LinearGLL.hpp:37-288 on
device vectors, calling libwavehip for every operator and vector kernel.

Same constructor arguments, init(), f0(), f1(), rk4() as the reference.  The
mesh/meshtags pair of the reference is replaced by a FunctionSpace plus the
two tagged boundary dof sets (tag 1 = Neumann source Gamma_1, tag 2 = first-order
absorbing Gamma_2) with their collocated facet masses; the FFCx form
L = c0^2 (g v ds(1) - 1/c0 v_n v ds(2)) (demo/cpu_planar3d/forms.ufl:19-24)
is applied in its diagonal GLL form by wf_boundary_apply.

Ghost exchange: `updater` (a VectorUpdater) supplies scatter_fwd / scatter_rev
where the reference calls la::Vector::scatter_fwd / scatter_rev(add)
(LinearGLL.hpp:110,127,164,176,284-285); None on one rank.  The forward update
of v inside f1 (LinearGLL.hpp:167) is not needed here: the boundary term is
applied by the owner of each boundary dof with the fully assembled facet mass
(distributed.owned_boundary), so no ghost value of v is ever read."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import la
from .operators import MassOperatorLumped, StiffnessOperator


def facet_lumped_mass(V, tag_of_face, tag: int):
    """Collocated facet mass m_Gamma[i] = sum_facets w_q |J_facet| for the box
    faces carrying `tag` (tag_of_face maps local face 2*axis+side -> tag).
    Host-side setup (numpy); returns (dof indices int32, masses float64)."""
    from .operators import tabulate_gll
    mesh = V.mesh
    p = V.degree
    n = p + 1
    pts, w, _ = tabulate_gll(p)
    nx, ny, nz = mesh.n
    NX, NY, NZ = V.lattice
    cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cid = cx + nx * (cy + ny * cz)
    m = {}
    acc = np.zeros(NX * NY * NZ)
    touched = np.zeros(NX * NY * NZ, dtype=bool)
    for axis, (cc, nn_) in enumerate(((cx, nx), (cy, ny), (cz, nz))):
        for side in (0, 1):
            if tag_of_face.get(2 * axis + side) != tag:
                continue
            cells = cid[cc == (0 if side == 0 else nn_ - 1)].reshape(-1)
            ta, tb = [d for d in range(3) if d != axis]
            bb, aa = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
            aa, bb = aa.reshape(-1), bb.reshape(-1)
            X = np.zeros((n * n, 3))
            X[:, axis] = float(side)
            X[:, ta] = pts[aa]
            X[:, tb] = pts[bb]
            # Q1 derivative table, vertex v = a + 2b + 4c
            dphi = np.zeros((3, n * n, 8))
            for v in range(8):
                bits = (v & 1, (v >> 1) & 1, (v >> 2) & 1)
                f = [X[:, d] if bits[d] else 1.0 - X[:, d] for d in range(3)]
                g = [np.ones(n * n) if bits[d] else -np.ones(n * n) for d in range(3)]
                dphi[0, :, v] = g[0] * f[1] * f[2]
                dphi[1, :, v] = f[0] * g[1] * f[2]
                dphi[2, :, v] = f[0] * f[1] * g[2]
            xc = mesh.x[mesh.geom_dofmap[cells]]
            J = np.einsum("fvi,jqv->fqij", xc, dphi)
            nrm = np.linalg.norm(np.cross(J[:, :, :, ta], J[:, :, :, tb]), axis=2)
            wq = (w[aa] * w[bb])[None, :] * nrm
            loc = np.zeros((n * n, 3), dtype=np.int64)
            loc[:, axis] = side * p
            loc[:, ta] = aa
            loc[:, tb] = bb
            ccx, ccy, ccz = cells % nx, (cells // nx) % ny, cells // (nx * ny)
            I = p * ccx[:, None] + loc[None, :, 0]
            Jd = p * ccy[:, None] + loc[None, :, 1]
            K = p * ccz[:, None] + loc[None, :, 2]
            dofs = (I + NX * (Jd + NY * K)).reshape(-1)
            np.add.at(acc, dofs, wq.reshape(-1))
            touched[dofs] = True
    idx = np.nonzero(touched)[0].astype(np.int32)
    return idx, acc[idx]


class LinearGLLOpt:
    def __init__(self, V, degreeOfBasis: int, speedOfSound: float, sourceFrequency: float,
                 pressureAmplitude: float, boundary=None, updater=None, device=None, structured=None,
                 tags=None):
        self.V = V
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.k_ = degreeOfBasis
        self.c0_ = speedOfSound
        self.freq0_ = sourceFrequency
        self.p0_ = pressureAmplitude
        self.w0_ = 2.0 * math.pi * self.freq0_
        self.T_ = 1.0 / self.freq0_
        self.alpha_ = 4.0
        self.updater = updater
        self.size_local = V.index_map.size_local
        N = V.ndofs
        z = lambda: torch.zeros(N, dtype=torch.float64, device=self.device)
        self.u, self.v, self.u_n, self.v_n = z(), z(), z(), z()
        self.m, self.b = z(), z()
        # LinearGLL.hpp:102-110: m = M 1, then scatter_rev(add)
        ones = torch.ones(N, dtype=torch.float64, device=self.device)
        self.mass_op = MassOperatorLumped(V, self.k_, structured=structured)
        self.mass_op(ones, self.m)
        if self.updater is not None:
            self.updater.scatter_rev(self.m)
            self.updater.scatter_fwd(self.m)   # keep ghost entries of m non-zero and consistent for b/m
        # LinearGLL.hpp:113-115: the boundary form
        if boundary is None:
            if tags is None:
                tags = {0: 1, 1: 2, 2: 2, 3: 2, 4: 2, 5: 2}
            if self.updater is not None:
                from .distributed import owned_boundary
                i1, m1 = owned_boundary(self.updater, V, tags, 1, self.device)
                i2, m2 = owned_boundary(self.updater, V, tags, 2, self.device)
            else:
                i1, m1 = facet_lumped_mass(V, tags, 1)
                i2, m2 = facet_lumped_mass(V, tags, 2)
        else:
            # boundary = ((idx1, m1), (idx2, m2)): rank-local sets (any mix of owned and ghost dofs, rank-local
            # facet masses).  On a partitioned mesh they go through the same accumulate-to-owner-and-filter
            # step as the tag-derived sets: the loop below applies the boundary term to owned dofs only and
            # never updates the ghosts of v (contract of owned_boundary_set).
            (i1, m1), (i2, m2) = boundary
            if self.updater is not None:
                from .distributed import owned_boundary_set
                i1, m1 = owned_boundary_set(self.updater, V, i1, m1, self.device)
                i2, m2 = owned_boundary_set(self.updater, V, i2, m2, self.device)
        td = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(self.device, dtype=dt)
        self.idx1, self.mG1 = td(i1, torch.int32), td(m1, torch.float64)
        self.idx2, self.mG2 = td(i2, torch.int32), td(m2, torch.float64)
        # LinearGLL.hpp:120-127
        self.stiff_op = StiffnessOperator(V, self.k_, {"c0": self.c0_}, structured=structured)
        # domain-decomposed run: interior cells overlap the forward ghost update
        self._split = False
        if self.updater is not None:
            self._split = self.stiff_op.set_ghost_faces(*[bool(v) for v in self.updater.part.owned_lo])
            if not self._split:   # arbitrary dofmap: split by the ghost positions the updater unpacks into
                self._split = self.stiff_op.set_ghost_dofs(self.updater.h_ghost_pos)
        self.stiff_op(self.u_n, self.b)
        if self.updater is not None:
            self.updater.scatter_rev(self.b)
        self.window_ = 0.0
        self.g_ = 0.0

    def init(self):
        self.u_n.zero_()
        self.v_n.zero_()

    def f0(self, t, u, v, result):
        la.copy(v, result)                                   # LinearGLL.hpp:141-144

    def f1(self, t, u, v, result):
        # LinearGLL.hpp:151-192
        if t < self.T_ * self.alpha_:
            self.window_ = 0.5 * (1.0 - math.cos(self.freq0_ * math.pi * t / self.alpha_))
        else:
            self.window_ = 1.0
        self.g_ = self.window_ * self.p0_ * self.w0_ / self.c0_ * math.cos(self.w0_ * t)
        if self._split:
            # same operations as below, reordered so that the cells that read no ghost
            # value run while the halo of u is in flight (update_fwd_begin/_end,
            # VectorUpdater.hpp:106-143)
            from .distributed import overlapped_apply
            la.fill(self.b, 0.0)
            overlapped_apply(self.stiff_op, self.updater, u, self.b)
            la.boundary_apply(self.idx1, self.mG1, self.c0_ ** 2 * self.g_,
                              self.idx2, self.mG2, -self.c0_, v, self.b)
            la.copy(u, self.u_n)
            la.copy(v, self.v_n)
            la.pointwise_div(self.b, self.m, result)
            return
        else:
            if self.updater is not None:
                self.updater.scatter_fwd(u)
            la.copy(u, self.u_n)
            la.copy(v, self.v_n)
            la.fill(self.b, 0.0)
            self.stiff_op(self.u_n, self.b)
        la.boundary_apply(self.idx1, self.mG1, self.c0_ ** 2 * self.g_,
                          self.idx2, self.mG2, -self.c0_, self.v_n, self.b)
        if self.updater is not None:
            self.updater.scatter_rev(self.b)
        la.pointwise_div(self.b, self.m, result)

    def rk4(self, startTime: float, finalTime: float, timeStep: float, max_steps: int | None = None):
        # LinearGLL.hpp:198-287
        t, tf, dt = startTime, finalTime, timeStep
        step = 0
        nl = self.size_local
        new = lambda: torch.zeros_like(self.u_n)
        u_, v_, un, vn, u0, v0, ku, kv = (new() for _ in range(8))
        la.copy(self.u_n, u_)
        la.copy(self.v_n, v_)
        la.copy(u_, ku)
        la.copy(v_, kv)
        a_runge = [0.0, 0.5, 0.5, 1.0]
        b_runge = [1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0]
        c_runge = [0.0, 0.5, 0.5, 1.0]
        while t < tf:
            dt = min(dt, tf - t)
            la.copy(u_, u0)
            la.copy(v_, v0)
            for i in range(4):
                la.copy(u0, un)
                la.copy(v0, vn)
                la.axpy(un, dt * a_runge[i], ku, un, nl)
                la.axpy(vn, dt * a_runge[i], kv, vn, nl)
                tn = t + c_runge[i] * dt
                self.f0(tn, un, vn, ku)
                self.f1(tn, un, vn, kv)
                la.axpy(u_, dt * b_runge[i], ku, u_, nl)
                la.axpy(v_, dt * b_runge[i], kv, v_, nl)
            t += dt
            step += 1
            if max_steps is not None and step >= max_steps:
                break
        la.copy(u_, self.u_n)
        la.copy(v_, self.v_n)
        if self.updater is not None:
            self.updater.scatter_fwd(self.u_n)
            self.updater.scatter_fwd(self.v_n)
        return t, step


def _rk4_fused(self, startTime: float, finalTime: float, timeStep: float, max_steps: int | None = None):
    """The same RK4 integration as rk4() (LinearGLL.hpp:198-287) with the vector
    algebra between two stiffness applies fused into one pass (wf_rk4_stage_bc) and
    the stage-0 copies removed by pointer rotation: per stage
    K (116 B/dof at P4) + 96 B/dof instead of K + 208 B/dof.  The diagonal boundary term
    of the NEXT right-hand side is left in b by the stage kernel (instead of zeros), so a
    stage is two launches: stiffness apply, fused vector algebra.  Same arithmetic
    expressions as the unfused loop (b receives the boundary term before the stiffness
    contributions instead of after them)."""
    from .distributed import overlapped_apply
    t, tf, dt = startTime, finalTime, timeStep
    step = 0
    new = lambda: torch.zeros_like(self.u_n)
    u0, v0 = self.u_n.clone(), self.v_n.clone()      # solution at the start of the step
    u_, v_ = new(), new()                              # running solution of the step
    un, vn_a, vn_b = new(), new(), new()
    b, m = self.b, self.m
    a_runge = [0.0, 0.5, 0.5, 1.0]
    b_runge = [1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0]
    c_runge = [0.0, 0.5, 0.5, 1.0]
    upd = self.updater

    def s1(tn):
        """c0^2 g(t): the coefficient of the Gamma_1 term (LinearGLL.hpp:153-162)."""
        if tn < self.T_ * self.alpha_:
            window = 0.5 * (1.0 - math.cos(self.freq0_ * math.pi * tn / self.alpha_))
        else:
            window = 1.0
        return self.c0_ ** 2 * window * self.p0_ * self.w0_ / self.c0_ * math.cos(self.w0_ * tn)

    def rhs(x_u):
        """b += K x_u (+ ghost updates); b holds the boundary term of this right-hand side on entry."""
        if self._split:
            # interior cells beside the halo of u, the interface cells and the reverse halo of b
            overlapped_apply(self.stiff_op, upd, x_u, b)
            return
        if upd is not None:
            upd.scatter_fwd(x_u)
        self.stiff_op(x_u, b)
        if upd is not None:
            upd.scatter_rev(b)

    if getattr(self, "_bc", None) is None:
        self._bc = la.BoundaryPlan(b.numel(), self.idx1.cpu().numpy(), self.mG1.cpu().numpy(), self.idx2.cpu().numpy(),
                                   self.mG2.cpu().numpy())
    # fold_boundary = False keeps the boundary term a launch of its own after each stage kernel (for comparison)
    fold = getattr(self, "fold_boundary", True)
    bc, s2 = self._bc, -self.c0_
    la.fill(b, 0.0)
    bc.apply(s1(t), s2, v0, b)           # boundary term of the first right-hand side; the stage kernels leave the others
    while t < tf:
        dt = min(dt, tf - t)
        # stage 0: un = u0, vn = v0 (a_0 = 0), read straight from u0 / v0
        x_u, x_v = u0, v0
        vn_next = vn_a
        for i in range(4):
            rhs(x_u)
            last = i == 3
            ur, vr = (u0, v0) if i == 0 else (u_, v_)
            if not last:
                la.rk4_stage(b, m, x_v, ur, vr, u_, v_, dt * b_runge[i], dt * a_runge[i + 1], u0, v0, un, vn_next,
                             bc=bc if fold else None, s1_next=s1(t + c_runge[i + 1] * dt), s2=s2)
                if not fold:
                    bc.apply(s1(t + c_runge[i + 1] * dt), s2, vn_next, b)
                x_u, x_v = un, vn_next
                vn_next = vn_b if vn_next is vn_a else vn_a
            else:
                # the next right-hand side is stage 0 of the next step: v = the updated solution, time t + dt
                # (after the last step b is left holding a boundary term nobody reads; f1 / rk4 zero b themselves)
                la.rk4_stage(b, m, x_v, ur, vr, u_, v_, dt * b_runge[i], bc=bc if fold else None, s1_next=s1(t + dt), s2=s2)
                if not fold:
                    bc.apply(s1(t + dt), s2, v_, b)
        u0, u_ = u_, u0          # the new solution becomes the next step's u0
        v0, v_ = v_, v0
        t += dt
        step += 1
        if max_steps is not None and step >= max_steps:
            break
    la.copy(u0, self.u_n)
    la.copy(v0, self.v_n)
    if upd is not None:
        upd.scatter_fwd(self.u_n)
        upd.scatter_fwd(self.v_n)
    return t, step


LinearGLLOpt.rk4_fused = _rk4_fused


def cfl_time_step(mesh, degree: int, c0: float, freq: float, CFL: float = 0.5):
    """demo/cpu_planar3d/main.cpp:48-66."""
    xc = mesh.x[mesh.geom_dofmap]
    d = np.linalg.norm(xc[:, :, None, :] - xc[:, None, :, :], axis=3)
    h = d.reshape(mesh.ncells, -1).max(axis=1).min()
    dt = CFL * h / (c0 * degree ** 2)
    period = 1.0 / freq
    stepPerPeriod = int(period / dt + 1)
    return period / stepPerPeriod, stepPerPeriod
