// Host-side tabulation (a1, a8, a13).  Stands in for the Basix
// calls of common/operators.hpp:16-24 and common/precompute.hpp:179-198: the
// GLL rule with P+1 points per direction and the GLL-warped Lagrange basis,
// restated from the published algorithms (Basix itself is not available
// offline).  Tensor ordering l = i + n*(j + n*k).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

namespace wf {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* last_error() { return g_error.c_str(); }

static double clamp101(double v)
{
  if (std::fabs(v + 1.0) <= 1e-8 + 1e-5) v = -1.0;
  if (std::fabs(v) <= 1e-8) v = 0.0;
  if (std::fabs(v - 1.0) <= 1e-8 + 1e-5) v = 1.0;
  return v;
}

// n-point Gauss-Lobatto-Legendre rule on [0,1], ascending.
void gll_points_weights(int n, double* pts, double* wts)
{
  const int N = n - 1;
  std::vector<double> x(n), w(n), P((size_t)n * n);
  for (int i = 0; i < n; ++i) x[i] = -std::cos(M_PI * i / N);
  auto legendre = [&](void) {
    for (int i = 0; i < n; ++i) {
      P[0 * n + i] = 1.0;
      P[1 * n + i] = x[i];
      for (int k = 2; k < n; ++k)
        P[k * n + i] = ((2 * k - 1) * x[i] * P[(k - 1) * n + i] - (k - 1) * P[(k - 2) * n + i]) / k;
    }
  };
  for (int it = 0; it < 100; ++it) {
    legendre();
    double dmax = 0.0;
    for (int i = 0; i < n; ++i) {
      const double dx = (x[i] * P[N * n + i] - P[(N - 1) * n + i]) / (n * P[N * n + i]);
      x[i] -= dx;
      dmax = std::fmax(dmax, std::fabs(dx));
    }
    if (dmax < 1e-16) break;
  }
  legendre();
  for (int i = 0; i < n; ++i) w[i] = 2.0 / (N * n * P[N * n + i] * P[N * n + i]);
  x[0] = -1.0;
  x[n - 1] = 1.0;
  for (int i = 0; i < n; ++i) {
    const double xs = 0.5 * (x[i] - x[n - 1 - i]);
    const double ws = 0.5 * (w[i] + w[n - 1 - i]);
    if (pts) pts[i] = 0.5 * (xs + 1.0);
    if (wts) wts[i] = 0.5 * ws;
  }
}

// m-point Gauss-Legendre (Gauss-Jacobi alpha = beta = 0) rule on [0,1], ascending:
// Newton on P_m from the Chebyshev guess, w = 2 / ((1 - x^2) P_m'(x)^2).
void gauss_legendre_points_weights(int m, double* pts, double* wts)
{
  for (int i = 0; i < (m + 1) / 2; ++i) {
    double x = -std::cos(M_PI * (i + 0.75) / (m + 0.5)), dp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p0 = 1.0, p1 = x;
      for (int k = 2; k <= m; ++k) {
        const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
        p0 = p1;
        p1 = p2;
      }
      if (m == 1) p0 = 1.0;
      dp = m * (x * p1 - p0) / (x * x - 1.0);
      const double dx = p1 / dp;
      x -= dx;
      if (std::fabs(dx) < 1e-16) break;
    }
    // one more evaluation of P_m' at the converged node
    double p0 = 1.0, p1 = x;
    for (int k = 2; k <= m; ++k) {
      const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
      p0 = p1;
      p1 = p2;
    }
    if (m == 1) p0 = 1.0;
    dp = m * (x * p1 - p0) / (x * x - 1.0);
    const double w = 2.0 / ((1.0 - x * x) * dp * dp);
    pts[i] = 0.5 * (x + 1.0);
    pts[m - 1 - i] = 0.5 * (1.0 - x);
    wts[i] = wts[m - 1 - i] = 0.5 * w;
  }
  if (m & 1) pts[m / 2] = 0.5;
}

// values / first derivatives of the Lagrange basis through `nodes` (n of them) at npts points, no clamp
static void lagrange_at(int n, const double* nodes, int npts, const double* pts, double* phi, double* dphi)
{
  for (int a = 0; a < n; ++a) {
    double denom = 1.0;
    for (int b = 0; b < n; ++b)
      if (b != a) denom *= nodes[a] - nodes[b];
    for (int q = 0; q < npts; ++q) {
      const double xq = pts[q];
      double num = 1.0;
      for (int b = 0; b < n; ++b)
        if (b != a) num *= xq - nodes[b];
      double s = 0.0;
      for (int c = 0; c < n; ++c) {
        if (c == a) continue;
        double t = 1.0;
        for (int b = 0; b < n; ++b)
          if (b != a && b != c) t *= xq - nodes[b];
        s += t;
      }
      if (phi) phi[q * n + a] = num / denom;
      if (dphi) dphi[q * n + a] = s / denom;
    }
  }
}

static void lagrange_1d(int n, const double* nodes, double* phi, double* dphi)
{
  for (int a = 0; a < n; ++a) {
    double denom = 1.0;
    for (int b = 0; b < n; ++b)
      if (b != a) denom *= nodes[a] - nodes[b];
    for (int q = 0; q < n; ++q) {
      const double xq = nodes[q];
      double num = 1.0;
      for (int b = 0; b < n; ++b)
        if (b != a) num *= xq - nodes[b];
      double s = 0.0;
      for (int c = 0; c < n; ++c) {
        if (c == a) continue;
        double t = 1.0;
        for (int b = 0; b < n; ++b)
          if (b != a && b != c) t *= xq - nodes[b];
        s += t;
      }
      if (phi) phi[q * n + a] = clamp101(num / denom);
      if (dphi) dphi[q * n + a] = clamp101(s / denom);
    }
  }
}

void gll_derivative_matrix(int P, double* D)
{
  const int n = P + 1;
  std::vector<double> pts(n);
  gll_points_weights(n, pts.data(), nullptr);
  lagrange_1d(n, pts.data(), nullptr, D);
}

}  // namespace wf

using namespace wf;

extern "C" {

const char* wf_last_error(void) { return wf::last_error(); }
const char* wf_version(void) { return "wavehip 0.1 (gfx950)"; }

int wf_tabulate_gll(int P, double* h_points, double* h_weights, double* h_D)
{
  if (P < 1 || P > kMaxDegree) {
    set_error("wf_tabulate_gll: degree must be 1..7");
    return WF_ERR_UNSUPPORTED;
  }
  const int n = P + 1;
  std::vector<double> pts(n), wts(n);
  gll_points_weights(n, pts.data(), wts.data());
  if (h_points) std::memcpy(h_points, pts.data(), n * sizeof(double));
  if (h_weights) std::memcpy(h_weights, wts.data(), n * sizeof(double));
  if (h_D) lagrange_1d(n, pts.data(), nullptr, h_D);
  return WF_OK;
}

int wf_quadrature_1d(int type, int degree, int* npts, double* h_points, double* h_weights)
{
  WF_REQUIRE(npts != nullptr, "wf_quadrature_1d: null output");
  WF_REQUIRE(degree >= 0, "wf_quadrature_1d: negative degree");
  // point counts of basix::quadrature::make_quadrature on the interval: Gauss-Jacobi
  // (degree + 2) / 2; GLL (degree + 4) / 2, i.e. P + 1 points for the reference's qdegree map
  // {2:3, 3:4, 4:6, 5:8, ...} (operators.hpp:63-72)
  int m;
  if (type == WF_QUAD_GLL)
    m = std::max(2, (degree + 4) / 2);
  else if (type == WF_QUAD_GAUSS_JACOBI)
    m = (degree + 2) / 2;
  else {
    set_error("wf_quadrature_1d: unknown rule");
    return WF_ERR_INVALID;
  }
  if (m > WF_MAX_QUAD_POINTS) {
    set_error("wf_quadrature_1d: more than WF_MAX_QUAD_POINTS points");
    return WF_ERR_UNSUPPORTED;
  }
  *npts = m;
  if (!h_points && !h_weights) return WF_OK;
  std::vector<double> p(m), w(m);
  if (type == WF_QUAD_GLL)
    gll_points_weights(m, p.data(), w.data());
  else
    gauss_legendre_points_weights(m, p.data(), w.data());
  if (h_points) std::memcpy(h_points, p.data(), m * sizeof(double));
  if (h_weights) std::memcpy(h_weights, w.data(), m * sizeof(double));
  return WF_OK;
}

int wf_tabulate_1d(int P, int variant, int npts, const double* h_points, int derivative, double* h_table)
{
  if (P < 1 || P > kMaxDegree) {
    set_error("wf_tabulate_1d: degree must be 1..7");
    return WF_ERR_UNSUPPORTED;
  }
  WF_REQUIRE(npts >= 1 && h_points && h_table, "wf_tabulate_1d: bad arguments");
  WF_REQUIRE(derivative == 0 || derivative == 1, "wf_tabulate_1d: derivative must be 0 or 1");
  const int n = P + 1;
  std::vector<double> nodes(n);
  if (variant == WF_VARIANT_GLL_WARPED)
    gll_points_weights(n, nodes.data(), nullptr);
  else if (variant == WF_VARIANT_EQUISPACED)
    for (int a = 0; a < n; ++a) nodes[a] = (double)a / P;
  else {
    set_error("wf_tabulate_1d: unknown Lagrange variant");
    return WF_ERR_INVALID;
  }
  lagrange_at(n, nodes.data(), npts, h_points, derivative == 0 ? h_table : nullptr, derivative == 1 ? h_table : nullptr);
  return WF_OK;
}

int wf_tabulate_dense(int P, double* h_table)
{
  if (P < 1 || P > kMaxDegree) {
    set_error("wf_tabulate_dense: degree must be 1..7");
    return WF_ERR_UNSUPPORTED;
  }
  WF_REQUIRE(h_table != nullptr, "wf_tabulate_dense: null output");
  const int n = P + 1, nd = n * n * n;
  std::vector<double> pts(n), phi(n * n), d(n * n);
  gll_points_weights(n, pts.data(), nullptr);
  lagrange_1d(n, pts.data(), phi.data(), d.data());
  for (int qk = 0; qk < n; ++qk)
    for (int qj = 0; qj < n; ++qj)
      for (int qi = 0; qi < n; ++qi) {
        const size_t q = qi + n * (qj + n * qk);
        for (int dk = 0; dk < n; ++dk)
          for (int dj = 0; dj < n; ++dj)
            for (int di = 0; di < n; ++di) {
              const size_t l = di + n * (dj + n * dk);
              const double px = phi[qi * n + di], py = phi[qj * n + dj], pz = phi[qk * n + dk];
              h_table[(0 * (size_t)nd + q) * nd + l] = clamp101(pz * py * px);
              h_table[(1 * (size_t)nd + q) * nd + l] = clamp101(pz * py * d[qi * n + di]);
              h_table[(2 * (size_t)nd + q) * nd + l] = clamp101(pz * d[qj * n + dj] * px);
              h_table[(3 * (size_t)nd + q) * nd + l] = clamp101(d[qk * n + dk] * py * px);
            }
      }
  return WF_OK;
}

int wf_reorder_dofmap(int ncells, int nd, const int32_t* h_perm, const int32_t* h_in, int32_t* h_out)
{
  WF_REQUIRE(ncells >= 0 && nd > 0 && h_perm && h_in && h_out, "wf_reorder_dofmap: bad arguments");
  for (int k = 0; k < nd; ++k) WF_REQUIRE(h_perm[k] >= 0 && h_perm[k] < nd, "wf_reorder_dofmap: perm out of range");
  for (size_t c = 0; c < (size_t)ncells; ++c)
    for (int k = 0; k < nd; ++k) h_out[c * nd + k] = h_in[c * nd + h_perm[k]];
  return WF_OK;
}

}  // extern "C"
