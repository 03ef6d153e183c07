// tsmm_demo -- demo/gpu_tsmm/main.cpp on MI355X: two tall-skinny products
// xq = xe . phi, ue = xq . phi with ncells = 100000, ndofs = 125 (column-major
// arrays, lda = ldc = ncells as in the reference), same output lines.
//   tsmm_demo [--ncells N] [--ndofs D] [--reps R]
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "wavehip.hpp"

int main(int argc, char* argv[])
{
  std::size_t ndofs = 125, ncells = 100000;
  int reps = 20;
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--ncells")) ncells = std::strtoull(argv[++i], nullptr, 10);
    else if (is("--ndofs")) ndofs = std::strtoull(argv[++i], nullptr, 10);
    else if (is("--reps")) reps = std::atoi(argv[++i]);
    else {
      std::cerr << "usage: tsmm_demo [--ncells N] [--ndofs D] [--reps R]\n";
      return 2;
    }
  }
  try {
    wavehip::set_device(0);
    wavehip::array<double> xe(ncells * ndofs), xq(ncells * ndofs), ue(ncells * ndofs), phi(ndofs * ndofs);
    wavehip::check(wf_fill((std::int64_t)(ncells * ndofs), 0.5, xe.data(), nullptr));   // the reference leaves these uninitialised
    wavehip::check(wf_fill((std::int64_t)(ndofs * ndofs), 0.01, phi.data(), nullptr));
    auto run = [&]() {
      wavehip::check(wf_tsmm(1, (std::int64_t)ncells, (int)ndofs, (int)ndofs, xe.data(), phi.data(), xq.data(), nullptr));
      wavehip::check(wf_tsmm(1, (std::int64_t)ncells, (int)ndofs, (int)ndofs, xq.data(), phi.data(), ue.data(), nullptr));
    };
    run();
    wavehip::check(wf_sync(nullptr));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run();
    wavehip::check(wf_sync(nullptr));
    double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::cout << "Number of cells: " << ncells;
    std::cout << "\nNumber of dofs: " << ndofs;
    std::cout << "\n#GFLOPs: " << (4.0 * ncells * ndofs * ndofs) / t / 1e9 << std::endl;   // main.cpp:58
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
