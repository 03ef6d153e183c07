// cg_demo -- the reference's matrix-free CG demo (demo/gpu_cg/main.cpp, BP1) on MI355X:
// solve M u = b with the mass operator of a degree-P hexahedral box mesh, where
// b = M f for f = x[0] + 4 (main.cpp:83 interpolates the same f), and report the iteration
// count like the reference ("its = N").  The solution must reproduce f.
//
//   cg_demo [--size N] [--degree P] [--op lumped|dense] [--kmax K] [--rtol R]
//
// --op dense: MassOperator with the GLL-warped basis and the Gauss rule of degree 2P
// (consistent mass, wavehip::MassOperator); --op lumped (default): the GLL-collocated mass
// the reference's demo builds (main.cpp:101-108: gll quadrature of degree 3 for P2).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "wavehip_box.hpp"

int main(int argc, char* argv[])
{
  int Nx = 16, degree = 2, kmax = 50;
  double rtol = 1e-4;   // main.cpp:118
  std::string opname = "lumped";
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--size")) Nx = std::atoi(argv[++i]);
    else if (is("--degree")) degree = std::atoi(argv[++i]);
    else if (is("--op")) opname = argv[++i];
    else if (is("--kmax")) kmax = std::atoi(argv[++i]);
    else if (is("--rtol")) rtol = std::atof(argv[++i]);
    else {
      std::cerr << "usage: cg_demo [--size N] [--degree P] [--op lumped|dense] [--kmax K] [--rtol R]\n";
      return 2;
    }
  }
  try {
    wavehip::set_device(0);
    auto mesh = wavehip::create_box({Nx, Nx, Nx});
    auto V = wavehip::create_functionspace(mesh, degree, /*build_dofmap=*/true);
    const std::int64_t N = V.ndofs();
    auto S = V.space();
    // f = x[0] + 4 at the dof coordinates (GLL nodes of the box lattice)
    std::vector<double> pts(degree + 1);
    wavehip::check(wf_tabulate_gll(degree, pts.data(), nullptr, nullptr));
    std::vector<double> f((std::size_t)N);
    const int NX = V.lattice[0], NY = V.lattice[1];
    for (std::int64_t g = 0; g < N; ++g) {
      const int I = (int)(g % NX);
      const int cx = I == NX - 1 ? Nx - 1 : I / degree, i = I - degree * cx;
      f[g] = (cx + pts[i]) / Nx + 4.0;
    }
    (void)NY;
    wavehip::array<double> df((std::size_t)N), b((std::size_t)N), u((std::size_t)N);
    df.set(f);
    wavehip::check(wf_fill(N, 0.0, b.data(), nullptr));
    wavehip::check(wf_fill(N, 0.0, u.data(), nullptr));
    std::unique_ptr<wavehip::MassOperator<double>> dense;
    std::unique_ptr<wavehip::MassOperatorLumped<double>> lumped;
    std::function<void(const double*, double*, void*)> matvec;
    if (opname == "dense") {
      dense = std::make_unique<wavehip::MassOperator<double>>(S, degree, WF_VARIANT_GLL_WARPED, WF_QUAD_GAUSS_JACOBI, 2 * degree);
      matvec = [&](const double* p, double* y, void* s) { dense->apply(p, y, s); };
    } else {
      lumped = std::make_unique<wavehip::MassOperatorLumped<double>>(S, degree);
      matvec = [&](const double* p, double* y, void* s) { lumped->apply(p, y, s); };
    }
    matvec(df.data(), b.data(), nullptr);   // b = M f  (the reference assembles L = inner(f, v) dx, main.cpp:86-93)
    double res = 0.0;
    const int its = wavehip::device::cg<double>(u.data(), b.data(), N, matvec, kmax, rtol, nullptr, nullptr, &res);
    wavehip::check(wf_sync(nullptr));
    auto hu = u.copy_to_host();
    double err = 0.0;
    for (std::int64_t g = 0; g < N; ++g) err = std::max(err, std::abs(hu[g] - f[g]));
    std::cout << "its = " << its << "\n";   // main.cpp:119
    std::cout << "relative residual: " << res << "\nmax |u - f|: " << err << "\nNumber of dofs: " << N << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
