// planar3d -- the reference's CPU demo (demo/cpu_planar3d/main.cpp) on MI355X:
// same physical constants, CFL rule and output lines.  The mesh is either read from an
// XDMF + HDF5 file with its facet tags, as the reference does (main.cpp:39-45:
// read_mesh(..., "planar3d"), read_meshtags(mesh, "planar3d_boundaries"); tag 1 = Gamma_1
// source, tag 2 = Gamma_2 absorbing) -- any conforming hexahedral mesh, one rank -- or a box
// generated in-process (the reference's own ../mesh.xdmf is not in its repository) with
// Gamma_1 = face x = 0 and Gamma_2 = every other face.
//
//   planar3d [--mesh FILE.xdmf [--grid NAME] [--tags NAME]]
//            [--size N] [--degree P] [--cfl C] [--steps S] [--length L] [--dump FILE]
//            [--periodic xyz] [--reference-order] [--markers]
//
// One process per GPU: with WORLD_SIZE > 1 in the environment (RANK, LOCAL_RANK,
// MASTER_PORT as torchrun exports them) --size is the number of cells per edge PER
// RANK of a Cartesian partition and the ghost exchange runs over RCCL
// (wavehip::VectorUpdater); the reference's mpirun ranks use DOLFINx's
// la::Vector::scatter_fwd/rev (common/LinearGLL.hpp:164-176).
// --periodic identifies opposite faces of the named axes (a single rank is then
// its own RCCL neighbour).  --reference-order runs LinearGLLOpt::rk4, the
// reference's unfused sequence of vector operations, instead of rk4_fused.
// --dump writes u_n then v_n of this rank (float64, local lattice order).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "wavehip_linear_gll.hpp"
#include "wavehip_mesh.hpp"

int main(int argc, char* argv[])
{
  int size = 18, degreeOfBasis = 4, nsteps_override = -1;
  double CFL = 0.5, domainLength = 0.1;
  const char* dump = nullptr;
  const char *mesh_file = nullptr, *grid_name = "planar3d", *tags_name = "planar3d_boundaries";
  std::array<bool, 3> periodic{false, false, false};
  bool reference_order = false, markers = false;
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--size")) size = std::atoi(argv[++i]);
    else if (is("--degree")) degreeOfBasis = std::atoi(argv[++i]);
    else if (is("--cfl")) CFL = std::atof(argv[++i]);
    else if (is("--steps")) nsteps_override = std::atoi(argv[++i]);
    else if (is("--length")) domainLength = std::atof(argv[++i]);
    else if (is("--dump")) dump = argv[++i];
    else if (is("--mesh")) mesh_file = argv[++i];
    else if (is("--grid")) grid_name = argv[++i];
    else if (is("--tags")) tags_name = argv[++i];
    else if (std::strcmp(argv[i], "--markers") == 0) markers = true;
    else if (is("--periodic")) {
      for (const char* c = argv[++i]; *c; ++c)
        if (*c >= 'x' && *c <= 'z') periodic[*c - 'x'] = true;
    } else if (std::strcmp(argv[i], "--reference-order") == 0) reference_order = true;
    else {
      std::cerr << "usage: planar3d [--mesh FILE.xdmf [--grid NAME] [--tags NAME]] [--size N] [--degree P] [--cfl C]"
                   " [--steps S] [--length L] [--dump FILE] [--periodic xyz] [--reference-order] [--markers]\n";
      return 2;
    }
  }
  try {
    std::cout.precision(15);
    const int world = wavehip::Comm::env_int("WORLD_SIZE", 1), rank = wavehip::Comm::env_int("RANK", 0);
    const bool exchange = world > 1 || periodic[0] || periodic[1] || periodic[2];
    std::unique_ptr<wavehip::Comm> comm;
    if (exchange)
      comm = wavehip::Comm::from_env();   // sets the device
    else
      wavehip::set_device(0);
    // Material / source parameters (demo/cpu_planar3d/main.cpp:25-33)
    double speedOfSound = 1500.0;
    double sourceFrequency = 0.5e6;
    double pressureAmplitude = 60000;
    double period = 1 / sourceFrequency;
    if (markers) wavehip::check(wf_markers_enable(1));   // roctx ranges (nvtxMarkA in demo/gpu_scatter_mpi/main.cpp:101-121)

    if (mesh_file) {
      // ---- the reference's path: mesh + facet tags from a file (main.cpp:39-45) ----
      if (exchange) throw std::runtime_error("--mesh runs on one rank");
      auto [mesh, tags] = wavehip::read_mesh(mesh_file, grid_name, tags_name);
      auto V = wavehip::create_functionspace(mesh, degreeOfBasis);
      auto [timeStepSize, stepPerPeriod] = wavehip::cfl_time_step(mesh, degreeOfBasis, speedOfSound, sourceFrequency, CFL);
      double startTime = 0.0;
      double finalTime = domainLength / speedOfSound + 8.0 / sourceFrequency;
      std::cout << "Number of step per period: " << stepPerPeriod << std::endl;
      std::cout << "dt = " << timeStepSize << std::endl;
      if (nsteps_override > 0) finalTime = nsteps_override * timeStepSize - 1e-13;
      int nstep = (int)((finalTime - startTime) / timeStepSize + 1);
      wavehip::LinearGLLOpt eqn(V.space(), wavehip::boundary_set(V, tags, 1), wavehip::boundary_set(V, tags, 2), degreeOfBasis,
                                speedOfSound, sourceFrequency, pressureAmplitude);
      std::cout << "Number of steps: " << nstep << std::endl;
      std::cout << "Degrees of freedom: " << V.ndofs << std::endl;
      eqn.init();
      auto t0 = std::chrono::steady_clock::now();
      int steps = reference_order ? eqn.rk4(startTime, finalTime, timeStepSize) : eqn.rk4_fused(startTime, finalTime, timeStepSize);
      wavehip::check(wf_sync(nullptr));
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
      return 0;
    }

    const auto procs = wavehip::decompose3d(world);
    const double L = domainLength;
    auto part = wavehip::create_distributed_box({size, size, size}, degreeOfBasis, world, rank, {0, 0, 0},
                                                {L * procs[0], L * procs[1], L * procs[2]}, periodic);
    const wavehip::BoxMesh& mesh = part->mesh;
    const wavehip::BoxSpace& V = part->V;
    std::unique_ptr<wavehip::VectorUpdater<double>> updater;
    if (exchange) updater = std::make_unique<wavehip::VectorUpdater<double>>(comm.get(), part->ghosts);

    // Temporal parameters (main.cpp:61-73)
    auto [timeStepSize, stepPerPeriod] = wavehip::cfl_time_step(mesh, degreeOfBasis, speedOfSound, sourceFrequency, CFL);
    double startTime = 0.0;
    double finalTime = domainLength / speedOfSound + 8.0 / sourceFrequency;
    (void)period;
    if (rank == 0) {
      std::cout << "Number of step per period: " << stepPerPeriod << std::endl;
      std::cout << "dt = " << timeStepSize << std::endl;
    }
    if (nsteps_override > 0) finalTime = nsteps_override * timeStepSize - 1e-13;
    int nstep = (int)((finalTime - startTime) / timeStepSize + 1);

    wavehip::LinearGLLOpt eqn(V, part->boundary_tags(), degreeOfBasis, speedOfSound, sourceFrequency, pressureAmplitude,
                              updater.get(), exchange ? part.get() : nullptr);
    if (rank == 0) {
      std::cout << "Number of steps: " << nstep << std::endl;
      std::cout << "Degrees of freedom: " << part->size_global << std::endl;
    }

    eqn.init();
    if (comm) comm->barrier();
    auto t0 = std::chrono::steady_clock::now();
    int steps = reference_order ? eqn.rk4(startTime, finalTime, timeStepSize) : eqn.rk4_fused(startTime, finalTime, timeStepSize);
    if (comm) comm->barrier();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0) {
      std::cout << "Steps taken: " << steps << std::endl;
      std::cout << "Solve time: " << secs << std::endl;
    }

    if (dump) {
      auto u = eqn.u_n->copy_to_host();
      auto v = eqn.v_n->copy_to_host();
      std::string name = dump;
      if (world > 1) name += "." + std::to_string(rank);
      FILE* f = std::fopen(name.c_str(), "wb");
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
