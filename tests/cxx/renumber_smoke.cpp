// Host-only check of the C++ mesh path's setup helpers (no GPU): a 4 x 3 x 5 box given as vertex array + cell
// list -> wavehip::create_functionspace (wf_fs_build) -> wavehip::renumber_lattice (wf_lattice_numbering).
// Prints "ndofs ok" when the new numbering is a permutation under which every cell's x-rows are mostly contiguous.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "wavehip_mesh.hpp"

int main()
{
  const int nx = 4, ny = 3, nz = 5, P = 3;
  wavehip::FileMesh m;
  for (int k = 0; k <= nz; ++k)
    for (int j = 0; j <= ny; ++j)
      for (int i = 0; i <= nx; ++i) {
        m.x.push_back(i / double(nx));
        m.x.push_back(j / double(ny));
        m.x.push_back(k / double(nz));
      }
  // cells in a scrambled order, so that the space's own (first-touch) numbering is not the lattice one
  std::vector<int> order;
  for (int c = 0; c < nx * ny * nz; ++c) order.push_back((c * 7) % (nx * ny * nz));
  for (int c : order) {
    const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
    for (int v = 0; v < 8; ++v)
      m.cells.push_back((i + (v & 1)) + (nx + 1) * ((j + ((v >> 1) & 1)) + (ny + 1) * (k + ((v >> 2) & 1))));
  }
  try {
    wavehip::MeshSpace V = wavehip::create_functionspace(m, P);
    const std::int64_t expect = (std::int64_t)(P * nx + 1) * (P * ny + 1) * (P * nz + 1);
    if (V.ndofs != expect) return 2;
    std::vector<std::int32_t> new_of_old = wavehip::renumber_lattice(V);
    std::vector<std::int32_t> sorted(new_of_old);
    std::sort(sorted.begin(), sorted.end());
    for (std::int64_t d = 0; d < V.ndofs; ++d)
      if (sorted[(std::size_t)d] != d) return 3;
    const int n = P + 1;
    long rows = 0, contiguous = 0;
    for (std::int64_t c = 0; c < m.ncells(); ++c)
      for (int kj = 0; kj < n * n; ++kj)
        for (int i = 0; i + 1 < n; ++i) {
          const std::int32_t a = V.dofmap[(std::size_t)c * n * n * n + kj * n + i], b = V.dofmap[(std::size_t)c * n * n * n + kj * n + i + 1];
          ++rows;
          contiguous += b == a + 1;
        }
    if (contiguous * 10 < rows * 9) return 4;
    std::printf("%lld ok\n", (long long)V.ndofs);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
