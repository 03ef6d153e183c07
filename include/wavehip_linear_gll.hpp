// wavehip_linear_gll.hpp -- C++ host driver of the explicit RK4 wave model:
// class LinearGLLOpt with the reference's member names and call sequence
// (common/LinearGLL.hpp:37-288), every vector on the device, every operation a
// libwavehip call.  The mesh/meshtags constructor arguments of the reference are
// replaced by a BoxSpace plus the facet tag map (tag 1 = Gamma_1 source,
// tag 2 = Gamma_2 absorbing); the FFCx form L (demo/cpu_planar3d/forms.ufl:19-24)
// is applied in its diagonal GLL form by wf_boundary_apply.
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
  std::unique_ptr<BoxStiffnessOperator<double>> stiff_op;

public:
  const BoxSpace& V;
  std::unique_ptr<array<double>> u_n, v_n;

  LinearGLLOpt(const BoxSpace& V_, const std::map<int, int>& facet_tags, int degreeOfBasis, double speedOfSound,
               double sourceFrequency, double pressureAmplitude)
      : V(V_)
  {
    k_ = degreeOfBasis;
    c0_ = speedOfSound;
    freq0_ = sourceFrequency;
    p0_ = pressureAmplitude;
    w0_ = 2.0 * M_PI * freq0_;
    T_ = 1.0 / freq0_;
    alpha_ = 4.0;
    N_ = V.ndofs();
    auto zeros = [&]() {
      auto a = std::make_unique<array<double>>((std::size_t)N_);
      check(wf_fill(N_, 0.0, a->data(), nullptr));
      return a;
    };
    u_n = zeros();
    v_n = zeros();
    m = zeros();
    b = zeros();
    // LinearGLL.hpp:102-110: m = M * 1
    {
      array<double> ones((std::size_t)N_);
      check(wf_fill(N_, 1.0, ones.data(), nullptr));
      wf_op* mass = nullptr;
      check(wf_op_create_box(WF_OP_MASS_LUMPED, k_, V.mesh->n[0], V.mesh->n[1], V.mesh->n[2], V.mesh->x.data(), 0.0,
                             WF_FLAG_NONE, &mass));
      check(wf_op_apply(mass, ones.data(), m->data(), nullptr));
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
    auto f1 = facet_lumped_mass(V, facet_tags, 1);
    auto f2 = facet_lumped_mass(V, facet_tags, 2);
    idx1 = up_i(f1.first);
    mG1 = up_d(f1.second);
    idx2 = up_i(f2.first);
    mG2 = up_d(f2.second);
    // LinearGLL.hpp:120-127
    stiff_op = std::make_unique<BoxStiffnessOperator<double>>(k_, V.mesh->n[0], V.mesh->n[1], V.mesh->n[2],
                                                              V.mesh->x.data(), c0_);
    stiff_op->apply(u_n->data(), b->data());
  }

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
    if (t < T_ * alpha_)
      window_ = 0.5 * (1.0 - std::cos(freq0_ * M_PI * t / alpha_));
    else
      window_ = 1.0;
    const double g = window_ * p0_ * w0_ / c0_ * std::cos(w0_ * t);
    kernels::copy(u, *u_n);   // scatter_fwd is the identity on one rank
    kernels::copy(v, *v_n);
    check(wf_fill(N_, 0.0, b->data(), nullptr));
    stiff_op->apply(u_n->data(), b->data());
    check(wf_boundary_apply((std::int32_t)idx1->size(), idx1->data(), mG1->data(), c0_ * c0_ * g,
                            (std::int32_t)idx2->size(), idx2->data(), mG2->data(), -c0_, v_n->data(), b->data(),
                            nullptr));
    check(wf_pointwise_div(N_, b->data(), m->data(), result.data(), nullptr));
  }

  /// Runge-Kutta 4th order solver (LinearGLL.hpp:198-287); returns the number of steps taken
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
    check(wf_sync(nullptr));
    return step;
  }
};

}  // namespace wavehip
