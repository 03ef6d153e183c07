// planar3d -- the reference's CPU demo (demo/cpu_planar3d/main.cpp) on MI355X:
// same physical constants, CFL rule and output lines; the mesh is a box generated
// in-process (the reference reads ../mesh.xdmf, which is not in its repository)
// with Gamma_1 = face x = 0 and Gamma_2 = every other face.
//
//   planar3d [--size N] [--degree P] [--cfl C] [--steps S] [--length L] [--dump FILE]
//
// --dump writes u_n then v_n (float64, lattice order) for the parity test.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "wavehip_linear_gll.hpp"

int main(int argc, char* argv[])
{
  int size = 18, degreeOfBasis = 4, nsteps_override = -1;
  double CFL = 0.5, domainLength = 0.1;
  const char* dump = nullptr;
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--size")) size = std::atoi(argv[++i]);
    else if (is("--degree")) degreeOfBasis = std::atoi(argv[++i]);
    else if (is("--cfl")) CFL = std::atof(argv[++i]);
    else if (is("--steps")) nsteps_override = std::atoi(argv[++i]);
    else if (is("--length")) domainLength = std::atof(argv[++i]);
    else if (is("--dump")) dump = argv[++i];
    else {
      std::cerr << "usage: planar3d [--size N] [--degree P] [--cfl C] [--steps S] [--length L] [--dump FILE]\n";
      return 2;
    }
  }
  try {
    std::cout.precision(15);
    wavehip::set_device(0);
    // Material / source parameters (demo/cpu_planar3d/main.cpp:25-33)
    double speedOfSound = 1500.0;
    double sourceFrequency = 0.5e6;
    double pressureAmplitude = 60000;
    double period = 1 / sourceFrequency;

    auto mesh = wavehip::create_box({size, size, size}, {0, 0, 0}, {domainLength, domainLength, domainLength});
    auto V = wavehip::create_functionspace(mesh, degreeOfBasis, /*build_dofmap=*/false);

    // Temporal parameters (main.cpp:61-73)
    auto [timeStepSize, stepPerPeriod] = wavehip::cfl_time_step(mesh, degreeOfBasis, speedOfSound, sourceFrequency, CFL);
    double startTime = 0.0;
    double finalTime = domainLength / speedOfSound + 8.0 / sourceFrequency;
    (void)period;
    std::cout << "Number of step per period: " << stepPerPeriod << std::endl;
    std::cout << "dt = " << timeStepSize << std::endl;
    if (nsteps_override > 0) finalTime = nsteps_override * timeStepSize - 1e-13;
    int nstep = (int)((finalTime - startTime) / timeStepSize + 1);

    std::map<int, int> tags{{0, 1}, {1, 2}, {2, 2}, {3, 2}, {4, 2}, {5, 2}};
    wavehip::LinearGLLOpt eqn(V, tags, degreeOfBasis, speedOfSound, sourceFrequency, pressureAmplitude);
    std::cout << "Number of steps: " << nstep << std::endl;
    std::cout << "Degrees of freedom: " << V.ndofs() << std::endl;

    eqn.init();
    auto t0 = std::chrono::steady_clock::now();
    int steps = eqn.rk4(startTime, finalTime, timeStepSize);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Steps taken: " << steps << std::endl;
    std::cout << "Solve time: " << secs << std::endl;

    if (dump) {
      auto u = eqn.u_n->copy_to_host();
      auto v = eqn.v_n->copy_to_host();
      FILE* f = std::fopen(dump, "wb");
      if (!f) throw std::runtime_error("cannot open dump file");
      std::fwrite(u.data(), sizeof(double), u.size(), f);
      std::fwrite(v.data(), sizeof(double), v.size(), f);
      std::fclose(f);
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
