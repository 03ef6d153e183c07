// operator_demo -- the reference's GPU operator demos on MI355X with the same
// flags and output lines: demo/gpu_operator_monolithic/main.cpp (mass, --op mass),
// demo/gpu_spectral_mass/main.cpp (--op spectral) and the stiffness operator the
// reference only has on the CPU (--op stiffness, default).
//
//   operator_demo [--size N] [--degree P] [--op stiffness|mass|spectral] [--reps R]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "wavehip_box.hpp"

int main(int argc, char* argv[])
{
  int Nx = 32, degree = 1, reps = 20;
  std::string opname = "stiffness";
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--size")) Nx = std::atoi(argv[++i]);
    else if (is("--degree")) degree = std::atoi(argv[++i]);
    else if (is("--op")) opname = argv[++i];
    else if (is("--reps")) reps = std::atoi(argv[++i]);
    else {
      std::cerr << "usage: operator_demo [--size N] [--degree P] [--op stiffness|mass|spectral] [--reps R]\n";
      return 2;
    }
  }
  try {
    wavehip::set_device(0);
    auto mesh = wavehip::create_box({Nx, Nx, Nx});
    auto V = wavehip::create_functionspace(mesh, degree, /*build_dofmap=*/false);
    const std::int64_t N = V.ndofs();
    wavehip::array<double> x((std::size_t)N), y((std::size_t)N);
    wavehip::check(wf_fill(N, 1.0, x.data(), nullptr));   // gpu_operator_monolithic/main.cpp:89
    wavehip::check(wf_fill(N, 0.0, y.data(), nullptr));

    wf_op* op = nullptr;
    const int kind = opname == "stiffness" ? WF_OP_STIFFNESS : WF_OP_MASS_LUMPED;
    const int flags = opname == "spectral" ? WF_FLAG_NO_FABS : WF_FLAG_NONE;
    wavehip::check(wf_op_create_box(kind, degree, Nx, Nx, Nx, mesh.x.data(), 1500.0, flags, &op));
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
    wf_op_destroy(op);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
