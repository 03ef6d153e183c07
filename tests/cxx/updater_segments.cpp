// wavehip::VectorUpdater<double> with several segments to one peer (the self peer of a one-rank
// communicator): five ncclSend / ncclRecv pairs in one group, one of them empty -- the code path
// cfg4's seven neighbours take (demo/gpu_scatter_mpi/VectorUpdater.hpp:106-208).  Run by
// tests/test_gpu_unstructured.py on the GPU box; prints "mismatches: N".
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "wavehip.hpp"

int main()
{
  using namespace wavehip;
  set_device(0);
  char id[WF_COMM_ID_BYTES];
  check(wf_comm_unique_id(id));
  Comm comm(id, 0, 1);
  const std::int32_t N = 20000;
  const std::vector<std::int32_t> seg = {1201, 0, 37, 1, 640};
  const std::int32_t total = std::accumulate(seg.begin(), seg.end(), 0);
  std::vector<std::int32_t> perm(N);
  std::iota(perm.begin(), perm.end(), 0);
  std::mt19937_64 rng(21);
  std::shuffle(perm.begin(), perm.end(), rng);
  GhostLists g;
  g.ndofs = N;
  for (std::size_t i = 0; i < seg.size(); ++i) {
    g.send_neighbors.push_back(0);
    g.recv_neighbors.push_back(0);
    g.send_offsets.push_back(g.send_offsets.back() + seg[i]);
    g.recv_offsets.push_back(g.recv_offsets.back() + seg[i]);
  }
  g.send_indices.assign(perm.begin(), perm.begin() + total);
  g.ghost_positions.assign(perm.begin() + total, perm.begin() + 2 * total);
  VectorUpdater<double> vu(&comm, g);

  std::vector<double> x0(N);
  for (auto& v : x0) v = (double)((std::int64_t)(rng() % 2001) - 1000);
  array<double> x(N), y(N);
  x.set(x0);
  y.set(x0);
  vu.update_fwd(x.data());
  vu.update_rev_begin(y.data());
  vu.update_rev_end(y.data());
  check(wf_sync(nullptr));
  std::vector<double> hx = x.copy_to_host(), hy = y.copy_to_host(), wx = x0, wy = x0;
  for (std::int32_t k = 0; k < total; ++k) {
    wx[g.ghost_positions[k]] = x0[g.send_indices[k]];
    wy[g.send_indices[k]] += x0[g.ghost_positions[k]];
  }
  long bad = 0;
  for (std::int32_t i = 0; i < N; ++i) bad += (hx[i] != wx[i]) + (hy[i] != wy[i]);
  std::printf("mismatches: %ld\n", bad);
  return bad == 0 ? 0 : 1;
}
