// Prints the ghost lists of wavehip::create_distributed_box (include/wavehip_box.hpp)
// so that the CPU test can compare them with wave_fenics_amd.distributed: the C++ and
// the Python host sides must name the same dofs in the same order, since ranks of
// either kind would exchange with each other.  Host-only (no GPU call).
//   partition_dump nx ny nz degree nproc rank px py pz   (p* = 0/1 periodic flags)
#include <cstdio>
#include <cstdlib>

#include "wavehip_box.hpp"

int main(int argc, char** argv)
{
  if (argc != 10) return 2;
  int a[9];
  for (int i = 0; i < 9; ++i) a[i] = std::atoi(argv[i + 1]);
  auto part = wavehip::create_distributed_box({a[0], a[1], a[2]}, a[3], a[4], a[5], {0, 0, 0}, {1, 1, 1},
                                              {a[6] != 0, a[7] != 0, a[8] != 0});
  const auto& g = part->ghosts;
  std::printf("procs %d %d %d coords %d %d %d owned_lo %d %d %d size_global %lld num_owned %lld\n", part->procs[0],
              part->procs[1], part->procs[2], part->coords[0], part->coords[1], part->coords[2], part->owned_lo[0],
              part->owned_lo[1], part->owned_lo[2], (long long)part->size_global, (long long)part->num_owned());
  for (std::size_t i = 0; i < g.send_neighbors.size(); ++i) {
    std::printf("send %d", g.send_neighbors[i]);
    for (std::int32_t k = g.send_offsets[i]; k < g.send_offsets[i + 1]; ++k) std::printf(" %d", g.send_indices[k]);
    std::printf("\n");
  }
  for (std::size_t i = 0; i < g.recv_neighbors.size(); ++i) {
    std::printf("recv %d", g.recv_neighbors[i]);
    for (std::int32_t k = g.recv_offsets[i]; k < g.recv_offsets[i + 1]; ++k) std::printf(" %d", g.ghost_positions[k]);
    std::printf("\n");
  }
  auto tags = part->boundary_tags();
  std::printf("tags");
  for (auto& kv : tags) std::printf(" %d:%d", kv.first, kv.second);
  std::printf("\n");
  return 0;
}
