// Prints the collocated facet masses of include/wavehip_box.hpp (facet_lumped_mass)
// on a box mesh whose vertex coordinates are read from a binary file, so that the
// CPU test can compare the C++ host code with the independent sympy derivation
// (tests/golden/independent.json).  Host-only (no GPU call).
//   facet_dump nx ny nz degree tag verts.bin
#include <cstdio>
#include <cstdlib>

#include "wavehip_box.hpp"

int main(int argc, char** argv)
{
  if (argc != 7) return 2;
  const int nx = std::atoi(argv[1]), ny = std::atoi(argv[2]), nz = std::atoi(argv[3]), p = std::atoi(argv[4]),
            tag = std::atoi(argv[5]);
  auto mesh = wavehip::create_box({nx, ny, nz});
  FILE* f = std::fopen(argv[6], "rb");
  if (!f || std::fread(mesh.x.data(), sizeof(double), mesh.x.size(), f) != mesh.x.size()) return 3;
  std::fclose(f);
  auto V = wavehip::create_functionspace(mesh, p, false);
  std::map<int, int> tags{{0, 1}, {1, 2}, {2, 2}, {3, 2}, {4, 2}, {5, 2}};
  auto fm = wavehip::facet_lumped_mass(V, tags, tag);
  for (std::size_t i = 0; i < fm.first.size(); ++i) std::printf("%d %.17g\n", fm.first[i], fm.second[i]);
  return 0;
}
