// scatter_demo -- the reference's ghost-update demo (demo/gpu_scatter_mpi/main.cpp)
// on MI355X: fill a distributed vector with the rank id, run VectorUpdater's
// forward and reverse updates over RCCL, check the ghosts and time the updates.
//
//   scatter_demo [--size N] [--degree P] [--reps R] [--periodic xyz]
//
// One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT from the
// launcher, as torchrun exports them); --size is cells per edge per rank.  With one
// rank, --periodic makes the rank its own neighbour so the exchange still runs.
// nvtxMarkA ranges of the reference (main.cpp:101-121) are not reproduced:
// rocprofv3 --kernel-trace names the pack/unpack kernels and the RCCL kernels.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "wavehip_box.hpp"

int main(int argc, char* argv[])
{
  int size = 32, degree = 2, reps = 50;
  std::array<bool, 3> periodic{false, false, false};
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0 && i + 1 < argc; };
    if (is("--size")) size = std::atoi(argv[++i]);
    else if (is("--degree")) degree = std::atoi(argv[++i]);
    else if (is("--reps")) reps = std::atoi(argv[++i]);
    else if (is("--periodic")) {
      for (const char* c = argv[++i]; *c; ++c)
        if (*c >= 'x' && *c <= 'z') periodic[*c - 'x'] = true;
    } else {
      std::cerr << "usage: scatter_demo [--size N] [--degree P] [--reps R] [--periodic xyz]\n";
      return 2;
    }
  }
  try {
    auto comm = wavehip::Comm::from_env();
    const int rank = comm->rank(), world = comm->size();
    auto part = wavehip::create_distributed_box({size, size, size}, degree, world, rank, {0, 0, 0}, {1, 1, 1}, periodic);
    const std::int64_t N = part->V.ndofs();
    wavehip::VectorUpdater<double> vu(comm.get(), part->ghosts);

    // x.set((double)mpi_rank)  (main.cpp:97)
    wavehip::array<double> x((std::size_t)N);
    wavehip::check(wf_fill(N, (double)rank, x.data(), nullptr));
    vu.update_fwd(x.data());
    wavehip::check(wf_sync(nullptr));
    auto h = x.copy_to_host();
    long bad = 0;
    const auto& g = part->ghosts;
    for (std::size_t i = 0; i < g.recv_neighbors.size(); ++i)
      for (std::int32_t k = g.recv_offsets[i]; k < g.recv_offsets[i + 1]; ++k)
        if (h[g.ghost_positions[k]] != (double)g.recv_neighbors[i]) ++bad;
    // reverse: every owner gains the ghost copies of its neighbours
    vu.update_rev(x.data());
    wavehip::check(wf_sync(nullptr));
    auto h2 = x.copy_to_host();
    std::vector<double> expect(h);
    for (std::size_t i = 0; i < g.send_neighbors.size(); ++i)
      for (std::int32_t k = g.send_offsets[i]; k < g.send_offsets[i + 1]; ++k) expect[g.send_indices[k]] += (double)rank;
    for (std::int64_t i = 0; i < N; ++i)
      if (h2[i] != expect[i]) ++bad;

    auto time = [&](auto&& fn) {
      comm->barrier();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < reps; ++i) fn();
      wavehip::check(wf_sync(nullptr));
      comm->barrier();
      return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    const double tf = time([&] { vu.update_fwd(x.data()); });
    const double tr = time([&] { vu.update_rev(x.data()); });
    if (rank == 0) {
      std::cout << "ranks: " << world << "  local dofs: " << N << "  send: " << g.send_indices.size()
                << "  recv: " << g.ghost_positions.size() << "  neighbours: " << g.send_neighbors.size() << "/"
                << g.recv_neighbors.size() << std::endl;
      std::cout << "update_fwd: " << tf * 1e6 << " us" << std::endl;
      std::cout << "update_rev: " << tr * 1e6 << " us" << std::endl;
    }
    std::cout << "rank " << rank << " mismatches: " << bad << std::endl;
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
}
