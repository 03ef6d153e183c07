// operator_demo -- the reference's GPU operator demos on MI355X with the same
// flags and output lines: demo/gpu_operator_monolithic/main.cpp (mass, --op mass),
// demo/gpu_spectral_mass/main.cpp (--op spectral) and the stiffness operator the
// reference only has on the CPU (--op stiffness, default).  --op dense is the
// MassOperator of common/cuda/mass.hpp (Phi^T D Phi) with the reference's element /
// quadrature arguments: --variant gll|equispaced, --quad gll|gauss, --qdegree Q
// (demo/gpu_operator: equispaced, gauss, 2P; gpu_operator_monolithic: gll, gll, P+1).
// --check (gpu_operator_monolithic/main.cpp:102-118): apply the lumped mass
// (MassOperatorCPU's role) to x = 1 as well and print every entry that differs by more
// than 1e-8 -- meaningful for the collocated rule (gll/gll), where both are the same
// diagonal; for other rules the sums over all dofs (= the volume) are compared.
//
//   operator_demo [--size N] [--degree P] [--op stiffness|mass|spectral|dense] [--reps R]
//                 [--variant gll|equispaced] [--quad gll|gauss] [--qdegree Q] [--check]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>

#include "wavehip_box.hpp"

int main(int argc, char* argv[])
{
  int Nx = 32, degree = 1, reps = 20, qdegree = -1;
  bool check = false;
  std::string opname = "stiffness", variant = "gll", quad = "gll";
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--size")) Nx = std::atoi(argv[++i]);
    else if (is("--degree")) degree = std::atoi(argv[++i]);
    else if (is("--op")) opname = argv[++i];
    else if (is("--reps")) reps = std::atoi(argv[++i]);
    else if (is("--variant")) variant = argv[++i];
    else if (is("--quad")) quad = argv[++i];
    else if (is("--qdegree")) qdegree = std::atoi(argv[++i]);
    else if (std::strcmp(argv[i], "--check") == 0) check = true;
    else {
      std::cerr << "usage: operator_demo [--size N] [--degree P] [--op stiffness|mass|spectral|dense] [--reps R]"
                   " [--variant gll|equispaced] [--quad gll|gauss] [--qdegree Q] [--check]\n";
      return 2;
    }
  }
  try {
    wavehip::set_device(0);
    auto mesh = wavehip::create_box({Nx, Nx, Nx});
    auto V = wavehip::create_functionspace(mesh, degree, /*build_dofmap=*/opname == "dense");
    const std::int64_t N = V.ndofs();
    wavehip::array<double> x((std::size_t)N), y((std::size_t)N);
    wavehip::check(wf_fill(N, 1.0, x.data(), nullptr));   // gpu_operator_monolithic/main.cpp:89
    wavehip::check(wf_fill(N, 0.0, y.data(), nullptr));

    wf_op* op = nullptr;
    std::unique_ptr<wavehip::MassOperator<double>> dense;
    if (opname == "dense") {
      // MassOperator<double> op(V, e, quad, qdegree)  (gpu_operator_monolithic/main.cpp:93-96)
      if (qdegree < 0) qdegree = (degree > 1) ? degree + 1 : degree;
      auto S = V.space();
      dense = std::make_unique<wavehip::MassOperator<double>>(
          S, degree, variant == "equispaced" ? WF_VARIANT_EQUISPACED : WF_VARIANT_GLL_WARPED,
          quad == "gauss" ? WF_QUAD_GAUSS_JACOBI : WF_QUAD_GLL, qdegree);
      op = dense->handle();
    } else {
      const int kind = opname == "stiffness" ? WF_OP_STIFFNESS : WF_OP_MASS_LUMPED;
      const int flags = opname == "spectral" ? WF_FLAG_NO_FABS : WF_FLAG_NONE;
      wavehip::check(wf_op_create_box(kind, degree, Nx, Nx, Nx, mesh.x.data(), 1500.0, flags, &op));
    }
    wf_op_info_t info{};
    wavehip::check(wf_op_info(op, &info));

    // one cold apply, timed like the reference (MPI_Wtime around op.apply)
    auto t0 = std::chrono::steady_clock::now();
    wavehip::check(wf_op_apply(op, x.data(), y.data(), nullptr));
    wavehip::check(wf_sync(nullptr));
    double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double xn = 0, yn = 0;
    {
      wavehip::array<double> r(1);
      std::vector<double> h;
      wavehip::check(wf_dot(N, x.data(), x.data(), r.data(), nullptr));
      xn = std::sqrt(r.copy_to_host()[0]);
      wavehip::check(wf_dot(N, y.data(), y.data(), r.data(), nullptr));
      yn = std::sqrt(r.copy_to_host()[0]);
    }
    if (check) {
      // the lumped (collocated) mass of the same space applied to x = 1
      wf_op* ref = nullptr;
      wavehip::check(wf_op_create_box(WF_OP_MASS_LUMPED, degree, Nx, Nx, Nx, mesh.x.data(), 0.0, WF_FLAG_NONE, &ref));
      wavehip::array<double> y1((std::size_t)N);
      wavehip::check(wf_fill(N, 0.0, y1.data(), nullptr));
      wavehip::check(wf_op_apply(ref, x.data(), y1.data(), nullptr));
      wavehip::check(wf_sync(nullptr));
      wf_op_destroy(ref);
      auto h = y.copy_to_host(), h1 = y1.copy_to_host();
      double s = 0, s1 = 0, n1 = 0;
      long bad = 0;
      for (std::int64_t i = 0; i < N; ++i) {
        s += h[i];
        s1 += h1[i];
        n1 += h1[i] * h1[i];
        if (std::abs(h1[i] - h[i]) > 1e-8) ++bad;
      }
      std::cout << "Y norm: " << std::sqrt(n1) << std::endl;
      std::cout << "check: sum(y) = " << s << "  sum(lumped) = " << s1 << "  entries differing by > 1e-8: " << bad
                << std::endl;
    }
    // warm repetitions
    wavehip::check(wf_sync(nullptr));
    auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) wavehip::check(wf_op_apply(op, x.data(), y.data(), nullptr));
    wavehip::check(wf_sync(nullptr));
    double tw = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() / std::max(reps, 1);

    std::cout << "X norm: " << xn << std::endl;
    std::cout << "Y norm: " << yn << std::endl;
    std::cout << "Number of cells: " << info.num_cells;
    std::cout << "\nNumber of dofs: " << info.num_dofs_cell;
    std::cout << "\nNumber of quads: " << info.num_quads;
    std::cout << "\n#Elapsed Time: " << t;
    std::cout << "\nDOF/s: " << N / t;
    std::cout << "\nDOF/s (warm, " << reps << " reps): " << N / tw;
    std::cout << "\nGB/s algorithmic (warm): " << info.alg_bytes / tw / 1e9 << std::endl;
    if (!dense) wf_op_destroy(op);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
