// wavehip_linear_gll.hpp -- C++ host driver of the explicit RK4 wave model:
// class LinearGLLOpt with the reference's member names and call sequence
// (common/LinearGLL.hpp:37-288), every vector on the device, every operation a
// libwavehip call.  The mesh/meshtags constructor arguments of the reference are
// replaced by either a BoxSpace plus the facet tag map (tag 1 = Gamma_1 source,
// tag 2 = Gamma_2 absorbing), or -- the mesh-FILE path of demo/cpu_planar3d/main.cpp:39-45 --
// a general Space plus the two tagged boundary dof sets (wavehip_mesh.hpp: read_mesh,
// create_functionspace, boundary_set); the FFCx form L (demo/cpu_planar3d/forms.ufl:19-24)
// is applied in its diagonal GLL form by wf_boundary_apply.
//
// Domain-decomposed runs pass a VectorUpdater (and the BoxPartition it was built
// from): scatter_fwd / scatter_rev(add) of LinearGLL.hpp:110,127,164,176,284-285
// become update_fwd / update_rev over RCCL, overlapped with the interior cells
// (wf_op_apply_overlapped).  The boundary term is applied by the owner of each
// boundary dof with the fully assembled facet mass, so the forward update of v
// (LinearGLL.hpp:167) is not needed.
#pragma once

#include <cmath>
#include <memory>

#include "wavehip_box.hpp"

namespace wavehip {

namespace kernels {
// kernels::copy / kernels::axpy of common/LinearGLL.hpp:15-35 on device arrays
inline void copy(const array<double>& in, array<double>& out) { check(wf_copy((std::int64_t)in.size(), in.data(), out.data(), nullptr)); }
inline void axpy(array<double>& r, double alpha, const array<double>& x, const array<double>& y, std::int64_t size_local)
{
  check(wf_axpy(size_local, alpha, x.data(), y.data(), r.data(), nullptr));
}
}  // namespace kernels

class LinearGLLOpt {
protected:
  int k_;         // degree of basis function
  double c0_;     // speed of sound (m/s)
  double freq0_;  // source frequency (Hz)
  double p0_;     // pressure amplitude (Pa)
  double w0_;     // angular frequency (rad/s)
  double T_;      // period (s)
  double alpha_;
  double window_ = 0.0;

  std::int64_t N_;  // vector length (owned + ghost), size_local on one rank
  std::unique_ptr<array<double>> m, b;
  std::unique_ptr<array<std::int32_t>> idx1, idx2;
  std::unique_ptr<array<double>> mG1, mG2;
  std::unique_ptr<detail::OpBase> stiff_op;
  wf_boundary* bc_ = nullptr;                  // boundary plan of the fused loop (created on first use)
  VectorUpdater<double>* updater_ = nullptr;   // nullptr on one rank
  bool split_ = false;                         // interior / interface overlap available

  double source(double t)
  {
    // LinearGLL.hpp:153-162
    if (t < T_ * alpha_)
      window_ = 0.5 * (1.0 - std::cos(freq0_ * M_PI * t / alpha_));
    else
      window_ = 1.0;
    return window_ * p0_ * w0_ / c0_ * std::cos(w0_ * t);
  }
  void boundary(double g, const double* v)
  {
    check(wf_boundary_apply((std::int32_t)idx1->size(), idx1->data(), mG1->data(), c0_ * c0_ * g,
                            (std::int32_t)idx2->size(), idx2->data(), mG2->data(), -c0_, v, b->data(), nullptr));
  }
  /// b += K u (+ ghost exchange on a partitioned mesh); b is zero on entry
  void stiffness(double* u)
  {
    if (split_) {
      check(wf_op_apply_overlapped(stiff_op->handle(), updater_->handle(), u, b->data(), nullptr));
      return;
    }
    if (updater_) updater_->update_fwd(u);
    stiff_op->apply(u, b->data());
    if (updater_) updater_->update_rev(b->data());
  }

  void set_parameters(int degreeOfBasis, double speedOfSound, double sourceFrequency, double pressureAmplitude)
  {
    k_ = degreeOfBasis;
    c0_ = speedOfSound;
    freq0_ = sourceFrequency;
    p0_ = pressureAmplitude;
    w0_ = 2.0 * M_PI * freq0_;
    T_ = 1.0 / freq0_;
    alpha_ = 4.0;
  }
  template <class C>
  static auto upload(const C& h)
  {
    auto a = std::make_unique<array<typename C::value_type>>(h.size());
    a->set(h);
    return a;
  }

public:
  const BoxSpace* V = nullptr;   // the box constructor's space (nullptr for the general-Space constructor)
  std::unique_ptr<array<double>> u_n, v_n;

  using BoundarySet = std::pair<std::vector<std::int32_t>, std::vector<double>>;

  /// The mesh-file path (demo/cpu_planar3d/main.cpp:39-45, common/LinearGLL.hpp:53-128): any conforming
  /// hexahedral space plus the dof sets of Gamma_1 (source) and Gamma_2 (absorbing) with their collocated
  /// facet masses (wavehip_mesh.hpp boundary_set).  One rank.
  LinearGLLOpt(const Space& S, const BoundarySet& gamma1, const BoundarySet& gamma2, int degreeOfBasis, double speedOfSound,
               double sourceFrequency, double pressureAmplitude)
  {
    set_parameters(degreeOfBasis, speedOfSound, sourceFrequency, pressureAmplitude);
    N_ = S.ndofs;
    auto zeros = [&]() {
      auto a = std::make_unique<array<double>>((std::size_t)N_);
      check(wf_fill(N_, 0.0, a->data(), nullptr));
      return a;
    };
    u_n = zeros();
    v_n = zeros();
    m = zeros();
    b = zeros();
    {   // LinearGLL.hpp:102-110: m = M * 1
      array<double> ones((std::size_t)N_);
      check(wf_fill(N_, 1.0, ones.data(), nullptr));
      MassOperatorLumped<double> mass(S, k_);
      mass.apply(ones.data(), m->data());
      check(wf_sync(nullptr));
    }
    idx1 = upload(gamma1.first);
    mG1 = upload(gamma1.second);
    idx2 = upload(gamma2.first);
    mG2 = upload(gamma2.second);
    std::map<std::string, double> params{{"c0", c0_}};
    stiff_op = std::make_unique<StiffnessOperator<double>>(S, k_, params);   // LinearGLL.hpp:120-127
    stiffness(u_n->data());
  }

  LinearGLLOpt(const BoxSpace& V_, const std::map<int, int>& facet_tags, int degreeOfBasis, double speedOfSound,
               double sourceFrequency, double pressureAmplitude, VectorUpdater<double>* updater = nullptr,
               const BoxPartition* part = nullptr)
      : updater_(updater), V(&V_)
  {
    if (updater && !part) throw std::runtime_error("LinearGLLOpt: a VectorUpdater needs its BoxPartition");
    set_parameters(degreeOfBasis, speedOfSound, sourceFrequency, pressureAmplitude);
    N_ = V->ndofs();
    auto zeros = [&]() {
      auto a = std::make_unique<array<double>>((std::size_t)N_);
      check(wf_fill(N_, 0.0, a->data(), nullptr));
      return a;
    };
    u_n = zeros();
    v_n = zeros();
    m = zeros();
    b = zeros();
    // LinearGLL.hpp:102-110: m = M * 1, scatter_rev(add)
    {
      array<double> ones((std::size_t)N_);
      check(wf_fill(N_, 1.0, ones.data(), nullptr));
      wf_op* mass = nullptr;
      check(wf_op_create_box(WF_OP_MASS_LUMPED, k_, V->mesh->n[0], V->mesh->n[1], V->mesh->n[2], V->mesh->x.data(), 0.0,
                             WF_FLAG_NONE, &mass));
      check(wf_op_apply(mass, ones.data(), m->data(), nullptr));
      if (updater_) {
        updater_->update_rev(m->data());
        updater_->update_fwd(m->data());   // ghost entries of m stay consistent (b / m runs over the whole array)
      }
      check(wf_sync(nullptr));
      wf_op_destroy(mass);
    }
    // LinearGLL.hpp:113-115: boundary form L
    auto up_i = [](const std::vector<std::int32_t>& h) {
      auto a = std::make_unique<array<std::int32_t>>(h.size());
      a->set(h);
      return a;
    };
    auto up_d = [](const std::vector<double>& h) {
      auto a = std::make_unique<array<double>>(h.size());
      a->set(h);
      return a;
    };
    auto facet = [&](int tag) {
      auto f = facet_lumped_mass(*V, facet_tags, tag);
      if (!updater_) return f;
      // accumulate the rank-local facet masses to their owners, keep owned dofs only
      std::vector<double> dense((std::size_t)N_, 0.0);
      for (std::size_t i = 0; i < f.first.size(); ++i) dense[f.first[i]] = f.second[i];
      array<double> d((std::size_t)N_);
      d.set(dense);
      updater_->update_rev(d.data());
      check(wf_sync(nullptr));
      dense = d.copy_to_host();
      std::pair<std::vector<std::int32_t>, std::vector<double>> out;
      for (std::int32_t i = 0; i < (std::int32_t)N_; ++i)
        if (dense[i] != 0.0 && part->owned(i)) {
          out.first.push_back(i);
          out.second.push_back(dense[i]);
        }
      return out;
    };
    auto f1 = facet(1);
    auto f2 = facet(2);
    idx1 = up_i(f1.first);
    mG1 = up_d(f1.second);
    idx2 = up_i(f2.first);
    mG2 = up_d(f2.second);
    // LinearGLL.hpp:120-127
    stiff_op = std::make_unique<BoxStiffnessOperator<double>>(k_, V->mesh->n[0], V->mesh->n[1], V->mesh->n[2],
                                                              V->mesh->x.data(), c0_);
    if (updater_) {
      const int rc = wf_op_set_ghost_faces(stiff_op->handle(), part->owned_lo[0], part->owned_lo[1], part->owned_lo[2]);
      if (rc != WF_OK && rc != WF_ERR_UNSUPPORTED) check(rc);
      split_ = rc == WF_OK;
    }
    stiffness(u_n->data());
  }

  ~LinearGLLOpt() { wf_boundary_destroy(bc_); }
  LinearGLLOpt(const LinearGLLOpt&) = delete;
  LinearGLLOpt& operator=(const LinearGLLOpt&) = delete;

  /// Set the initial values of u and v (LinearGLL.hpp:131-134)
  void init()
  {
    check(wf_fill(N_, 0.0, u_n->data(), nullptr));
    check(wf_fill(N_, 0.0, v_n->data(), nullptr));
  }

  /// du/dt = f0(t, u, v)  (LinearGLL.hpp:141-144)
  void f0(double& /*t*/, array<double>& /*u*/, array<double>& v, array<double>& result) { kernels::copy(v, result); }

  /// dv/dt = f1(t, u, v)  (LinearGLL.hpp:151-192)
  void f1(double& t, array<double>& u, array<double>& v, array<double>& result)
  {
    const double g = source(t);
    check(wf_fill(N_, 0.0, b->data(), nullptr));
    stiffness(u.data());        // scatter_fwd(u); K; scatter_rev(b)
    kernels::copy(u, *u_n);
    kernels::copy(v, *v_n);
    boundary(g, v_n->data());
    check(wf_pointwise_div(N_, b->data(), m->data(), result.data(), nullptr));
  }

  /// Runge-Kutta 4th order solver (LinearGLL.hpp:198-287), the reference's sequence of
  /// vector operations; returns the number of steps taken
  int rk4(double& startTime, double& finalTime, double& timeStep)
  {
    double t = startTime, tf = finalTime, dt = timeStep;
    int step = 0;
    auto mk = [&]() { return std::make_unique<array<double>>((std::size_t)N_); };
    auto u_ = mk(), v_ = mk(), un = mk(), vn = mk(), u0 = mk(), v0 = mk(), ku = mk(), kv = mk();
    kernels::copy(*u_n, *u_);
    kernels::copy(*v_n, *v_);
    kernels::copy(*u_, *ku);
    kernels::copy(*v_, *kv);
    const int n_RK = 4;
    const double a_runge[4] = {0.0, 0.5, 0.5, 1.0};
    const double b_runge[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
    const double c_runge[4] = {0.0, 0.5, 0.5, 1.0};
    double tn;
    while (t < tf) {
      dt = std::min(dt, tf - t);
      kernels::copy(*u_, *u0);
      kernels::copy(*v_, *v0);
      for (int i = 0; i < n_RK; i++) {
        kernels::copy(*u0, *un);
        kernels::copy(*v0, *vn);
        kernels::axpy(*un, dt * a_runge[i], *ku, *un, N_);
        kernels::axpy(*vn, dt * a_runge[i], *kv, *vn, N_);
        tn = t + c_runge[i] * dt;
        f0(tn, *un, *vn, *ku);
        f1(tn, *un, *vn, *kv);
        kernels::axpy(*u_, dt * b_runge[i], *ku, *u_, N_);
        kernels::axpy(*v_, dt * b_runge[i], *kv, *v_, N_);
      }
      t += dt;
      step += 1;
    }
    kernels::copy(*u_, *u_n);
    kernels::copy(*v_, *v_n);
    finish();
    return step;
  }

  /// The same integration with the vector algebra between two stiffness applies fused
  /// into ONE pass (wf_rk4_stage: divide, f0, both solution updates, the next stage's
  /// un / vn and the zeroing of b) and the stage-0 copies removed by pointer rotation.
  /// Same arithmetic expressions as rk4(); 72-96 B/dof of vector traffic per stage
  /// instead of 208.
  int rk4_fused(double& startTime, double& finalTime, double& timeStep)
  {
    double t = startTime, tf = finalTime, dt = timeStep;
    int step = 0;
    auto mk = [&]() { return std::make_unique<array<double>>((std::size_t)N_); };
    auto u0 = mk(), v0 = mk(), u_ = mk(), v_ = mk(), un = mk(), vn_a = mk(), vn_b = mk();
    kernels::copy(*u_n, *u0);
    kernels::copy(*v_n, *v0);
    const double a_runge[4] = {0.0, 0.5, 0.5, 1.0};
    const double b_runge[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
    const double c_runge[4] = {0.0, 0.5, 0.5, 1.0};
    // The boundary term of each right-hand side is left in b by the stage kernel in front of it
    // (wf_rk4_stage_bc), the first one by a plain launch: a stage is two launches, stiffness + vector algebra.
    if (!bc_) {
      const auto i1 = idx1->copy_to_host(), i2 = idx2->copy_to_host();
      const auto a1 = mG1->copy_to_host(), a2 = mG2->copy_to_host();
      check(wf_boundary_create(N_, (std::int32_t)i1.size(), i1.data(), a1.data(), (std::int32_t)i2.size(), i2.data(), a2.data(), &bc_));
    }
    check(wf_fill(N_, 0.0, b->data(), nullptr));
    check(wf_boundary_apply_plan(bc_, c0_ * c0_ * source(t), -c0_, v0->data(), b->data(), nullptr));
    while (t < tf) {
      dt = std::min(dt, tf - t);
      double* x_u = u0->data();           // stage 0 reads u0 / v0 directly (a_0 = 0)
      double* x_v = v0->data();
      double* vn_next = vn_a->data();
      for (int i = 0; i < 4; ++i) {
        stiffness(x_u);
        const double* ur = i == 0 ? u0->data() : u_->data();
        const double* vr = i == 0 ? v0->data() : v_->data();
        const int has_next = i < 3;
        // time of the next right-hand side: the next stage, or stage 0 of the next step
        const double g_next = source(has_next ? t + c_runge[i + 1] * dt : t + dt);
        check(wf_rk4_stage_bc(N_, dt * b_runge[i], has_next ? dt * a_runge[i + 1] : 0.0, has_next, b->data(), m->data(), x_v,
                              ur, vr, u_->data(), v_->data(), u0->data(), v0->data(), un->data(), vn_next, bc_,
                              c0_ * c0_ * g_next, -c0_, nullptr));
        if (has_next) {
          x_u = un->data();
          x_v = vn_next;
          vn_next = vn_next == vn_a->data() ? vn_b->data() : vn_a->data();
        }
      }
      std::swap(u0, u_);   // the new solution is the next step's u0
      std::swap(v0, v_);
      t += dt;
      step += 1;
    }
    kernels::copy(*u0, *u_n);
    kernels::copy(*v0, *v_n);
    finish();
    return step;
  }

protected:
  void finish()
  {
    if (updater_) {   // LinearGLL.hpp:284-285
      updater_->update_fwd(u_n->data());
      updater_->update_fwd(v_n->data());
    }
    check(wf_sync(nullptr));
  }
};

}  // namespace wavehip
