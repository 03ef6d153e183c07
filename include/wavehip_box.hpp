// wavehip_box.hpp -- host-side box mesh / function space in C++ (the arrays a
// dolfinx::mesh::create_box + fem::create_functionspace pair would provide,
// demo/gpu_operator/main.cpp:60-72), with this engine's lexicographic numbering,
// and the collocated facet masses of the boundary form
// (demo/cpu_planar3d/forms.ufl:19-24).
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <functional>
#include <cstdint>
#include <map>
#include <vector>

#include "wavehip.hpp"

namespace wavehip {

struct BoxMesh {
  std::array<int, 3> n{};                    // cells per direction
  std::vector<double> x;                     // [nverts][3]
  std::vector<std::int32_t> geom_dofmap;     // [ncells][8], vertex v = a + 2b + 4c
  std::int32_t ncells() const { return n[0] * n[1] * n[2]; }
  std::int32_t nverts() const { return (n[0] + 1) * (n[1] + 1) * (n[2] + 1); }
};

/// mesh::create_box(comm, {lo, hi}, {nx, ny, nz}, hexahedron)
inline BoxMesh create_box(std::array<int, 3> n, std::array<double, 3> lo = {0, 0, 0},
                          std::array<double, 3> hi = {1, 1, 1})
{
  BoxMesh m;
  m.n = n;
  const int nx = n[0], ny = n[1], nz = n[2];
  m.x.resize((std::size_t)m.nverts() * 3);
  for (int c = 0; c <= nz; ++c)
    for (int b = 0; b <= ny; ++b)
      for (int a = 0; a <= nx; ++a) {
        const std::size_t v = a + (std::size_t)(nx + 1) * (b + (std::size_t)(ny + 1) * c);
        // numpy.linspace(lo, hi, n+1): lo + step*i, last point exactly hi
        m.x[v * 3 + 0] = a == nx ? hi[0] : lo[0] + (hi[0] - lo[0]) / nx * a;
        m.x[v * 3 + 1] = b == ny ? hi[1] : lo[1] + (hi[1] - lo[1]) / ny * b;
        m.x[v * 3 + 2] = c == nz ? hi[2] : lo[2] + (hi[2] - lo[2]) / nz * c;
      }
  m.geom_dofmap.resize((std::size_t)m.ncells() * 8);
  for (int cz = 0; cz < nz; ++cz)
    for (int cy = 0; cy < ny; ++cy)
      for (int cx = 0; cx < nx; ++cx) {
        const std::size_t cell = cx + (std::size_t)nx * (cy + (std::size_t)ny * cz);
        for (int v = 0; v < 8; ++v)
          m.geom_dofmap[cell * 8 + v]
              = (cx + (v & 1)) + (nx + 1) * ((cy + ((v >> 1) & 1)) + (ny + 1) * (cz + ((v >> 2) & 1)));
      }
  return m;
}

/// The data of fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped)).
struct BoxSpace {
  const BoxMesh* mesh = nullptr;
  int degree = 0;
  std::array<int, 3> lattice{};              // (NX, NY, NZ)
  std::vector<std::int32_t> dofmap;          // [ncells][nd], tensor order (empty if !build_dofmap)
  std::int32_t ndofs() const { return lattice[0] * lattice[1] * lattice[2]; }

  Space space() const
  {
    Space S;
    S.degree = degree;
    S.ncells = mesh->ncells();
    S.ndofs = ndofs();
    S.dofmap = dofmap.data();
    S.nverts = mesh->nverts();
    S.x = mesh->x.data();
    S.geom_dofmap = mesh->geom_dofmap.data();
    return S;
  }
};

inline BoxSpace create_functionspace(const BoxMesh& mesh, int degree, bool build_dofmap = true)
{
  BoxSpace V;
  V.mesh = &mesh;
  V.degree = degree;
  const int p = degree, n = p + 1, nd = n * n * n;
  const int nx = mesh.n[0], ny = mesh.n[1], nz = mesh.n[2];
  V.lattice = {p * nx + 1, p * ny + 1, p * nz + 1};
  if (build_dofmap) {
    V.dofmap.resize((std::size_t)mesh.ncells() * nd);
    const std::int64_t NX = V.lattice[0], NY = V.lattice[1];
    for (int cz = 0; cz < nz; ++cz)
      for (int cy = 0; cy < ny; ++cy)
        for (int cx = 0; cx < nx; ++cx) {
          const std::size_t cell = cx + (std::size_t)nx * (cy + (std::size_t)ny * cz);
          for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i)
                V.dofmap[cell * nd + i + n * (j + n * k)]
                    = (std::int32_t)((p * cx + i) + NX * ((p * cy + j) + NY * (p * cz + k)));
        }
  }
  return V;
}

/// Collocated facet mass of the box faces carrying `tag`: m[i] = sum w_q |J_facet|
/// (the diagonal GLL form of inner(g, v) * ds(tag), forms.ufl:19-24).
/// tag_of_face: local face 2*axis + side -> tag.  Returns (dof indices, masses).
inline std::pair<std::vector<std::int32_t>, std::vector<double>>
facet_lumped_mass(const BoxSpace& V, const std::map<int, int>& tag_of_face, int tag)
{
  const BoxMesh& mesh = *V.mesh;
  const int p = V.degree, n = p + 1;
  std::vector<double> pts(n), wts(n);
  check(wf_tabulate_gll(p, pts.data(), wts.data(), nullptr));
  const std::int64_t NX = V.lattice[0], NY = V.lattice[1];
  std::map<std::int32_t, double> acc;
  const int nn[3] = {mesh.n[0], mesh.n[1], mesh.n[2]};
  for (int axis = 0; axis < 3; ++axis)
    for (int side = 0; side < 2; ++side) {
      auto it = tag_of_face.find(2 * axis + side);
      if (it == tag_of_face.end() || it->second != tag) continue;
      const int ta = axis == 0 ? 1 : 0, tb = axis == 2 ? 1 : 2;
      int c[3];
      c[axis] = side == 0 ? 0 : nn[axis] - 1;
      for (c[tb] = 0; c[tb] < nn[tb]; ++c[tb])
        for (c[ta] = 0; c[ta] < nn[ta]; ++c[ta]) {
          const std::size_t cell = c[0] + (std::size_t)nn[0] * (c[1] + (std::size_t)nn[1] * c[2]);
          double xv[8][3];
          for (int v = 0; v < 8; ++v)
            for (int d = 0; d < 3; ++d) xv[v][d] = mesh.x[(std::size_t)mesh.geom_dofmap[cell * 8 + v] * 3 + d];
          for (int b = 0; b < n; ++b)
            for (int a = 0; a < n; ++a) {
              double X[3];
              X[axis] = side;
              X[ta] = pts[a];
              X[tb] = pts[b];
              // tangents dx/dX_ta, dx/dX_tb of the trilinear map
              double t1[3] = {0, 0, 0}, t2[3] = {0, 0, 0};
              for (int v = 0; v < 8; ++v) {
                const int bits[3] = {v & 1, (v >> 1) & 1, (v >> 2) & 1};
                double f[3], g[3];
                for (int d = 0; d < 3; ++d) {
                  f[d] = bits[d] ? X[d] : 1.0 - X[d];
                  g[d] = bits[d] ? 1.0 : -1.0;
                }
                double da = g[ta], db = g[tb];
                for (int d = 0; d < 3; ++d) {
                  if (d != ta) da *= f[d];
                  if (d != tb) db *= f[d];
                }
                for (int d = 0; d < 3; ++d) {
                  t1[d] += xv[v][d] * da;
                  t2[d] += xv[v][d] * db;
                }
              }
              const double cx = t1[1] * t2[2] - t1[2] * t2[1], cy = t1[2] * t2[0] - t1[0] * t2[2],
                           cz = t1[0] * t2[1] - t1[1] * t2[0];
              const double wq = wts[a] * wts[b] * std::sqrt(cx * cx + cy * cy + cz * cz);
              int loc[3];
              loc[axis] = side * p;
              loc[ta] = a;
              loc[tb] = b;
              const std::int32_t dof = (std::int32_t)((p * c[0] + loc[0]) + NX * ((p * c[1] + loc[1]) + NY * (p * c[2] + loc[2])));
              acc[dof] += wq;
            }
        }
    }
  std::pair<std::vector<std::int32_t>, std::vector<double>> out;
  for (auto& kv : acc) {
    out.first.push_back(kv.first);
    out.second.push_back(kv.second);
  }
  return out;
}

// ---- domain decomposition (idea of demo/gpu_cg/mesh.hpp:37-63) -------
/// Split nproc into px >= py >= pz, as balanced as possible (1, 2x1x1, 2x2x1, 2x2x2).
inline std::array<int, 3> decompose3d(int nproc)
{
  std::array<int, 3> best{nproc, 1, 1};
  long best_score = -1;
  for (int a = 1; a <= nproc; ++a) {
    if (nproc % a) continue;
    for (int b = 1; b <= nproc / a; ++b) {
      if ((nproc / a) % b) continue;
      std::array<int, 3> d{a, b, nproc / a / b};
      std::sort(d.begin(), d.end(), std::greater<int>());
      const long score = (long)(d[0] - d[2]) * 100000 + d[0];
      if (best_score < 0 || score < best_score) {
        best_score = score;
        best = d;
      }
    }
  }
  return best;
}
/// rank = rz + pz*(ry + py*rx)  (compute_cartesian_indices, z fastest)
inline std::array<int, 3> rank_coords(int rank, const std::array<int, 3>& procs)
{
  return {rank / (procs[1] * procs[2]), (rank / procs[2]) % procs[1], rank % procs[2]};
}
inline int coords_rank(const std::array<int, 3>& c, const std::array<int, 3>& procs)
{
  return c[2] + procs[2] * (c[1] + procs[1] * c[0]);
}

/// One rank's part of a Cartesian partition of the box: local mesh and space, the
/// ghost lists for VectorUpdater, the owned/ghost split.  A lattice point shared by
/// several ranks belongs to the lowest one, so the ghosts of a rank are its lower
/// lattice planes; local vectors keep the lattice numbering (ghosts interleaved).
struct BoxPartition {
  std::array<int, 3> procs{1, 1, 1}, coords{0, 0, 0}, n_local{};
  std::array<bool, 3> periodic{false, false, false};
  std::array<int, 3> owned_lo{0, 0, 0};       // first owned lattice index per axis
  int rank = 0, degree = 0;
  std::int64_t size_global = 0;
  BoxMesh mesh;
  BoxSpace V;
  GhostLists ghosts;

  std::int64_t num_owned() const
  {
    return (std::int64_t)(V.lattice[0] - owned_lo[0]) * (V.lattice[1] - owned_lo[1]) * (V.lattice[2] - owned_lo[2]);
  }
  bool owned(std::int32_t dof) const
  {
    const int I = dof % V.lattice[0], J = (dof / V.lattice[0]) % V.lattice[1], K = dof / (V.lattice[0] * V.lattice[1]);
    return I >= owned_lo[0] && J >= owned_lo[1] && K >= owned_lo[2];
  }
  /// facet tags under the cfg1 convention: global face x = lo -> 1, other global faces -> 2
  std::map<int, int> boundary_tags() const
  {
    std::map<int, int> tags;
    for (int a = 0; a < 3; ++a) {
      if (periodic[a]) continue;
      if (coords[a] == 0) tags[2 * a] = a == 0 ? 1 : 2;
      if (coords[a] == procs[a] - 1) tags[2 * a + 1] = 2;
    }
    return tags;
  }
};

/// Weak-scaled box: n cells per direction PER RANK, global mesh (px nx, py ny, pz nz)
/// cells on [lo, hi].  periodic[a] identifies the upper face of axis a with the lower
/// one (a rank can then be its own neighbour).  Returned by pointer: V refers to mesh.
inline std::unique_ptr<BoxPartition> create_distributed_box(std::array<int, 3> n, int degree, int nproc, int rank,
                                                            std::array<double, 3> lo = {0, 0, 0},
                                                            std::array<double, 3> hi = {1, 1, 1},
                                                            std::array<bool, 3> periodic = {false, false, false})
{
  auto part = std::make_unique<BoxPartition>();
  BoxPartition& P = *part;
  P.procs = decompose3d(nproc);
  P.coords = rank_coords(rank, P.procs);
  P.rank = rank;
  P.degree = degree;
  P.n_local = n;
  P.periodic = periodic;
  std::array<double, 3> llo, lhi;
  P.size_global = 1;
  for (int a = 0; a < 3; ++a) {
    const int gn = P.procs[a] * n[a];
    const double h = (hi[a] - lo[a]) / gn;
    llo[a] = lo[a] + h * n[a] * P.coords[a];
    lhi[a] = lo[a] + h * n[a] * (P.coords[a] + 1);
    P.owned_lo[a] = (P.coords[a] > 0 || periodic[a]) ? 1 : 0;
    P.size_global *= (std::int64_t)degree * gn + (periodic[a] ? 0 : 1);
  }
  P.mesh = create_box(n, llo, lhi);
  P.V = create_functionspace(P.mesh, degree, /*build_dofmap=*/false);
  const int NX = P.V.lattice[0], NY = P.V.lattice[1], NZ = P.V.lattice[2];
  const int hi_idx[3] = {NX - 1, NY - 1, NZ - 1};
  // kind -1: lower plane, +1: upper plane, 0: the owned range of that axis
  auto lattice_indices = [&](const int kinds[3]) {
    int b[3], e[3];
    for (int a = 0; a < 3; ++a) {
      b[a] = kinds[a] < 0 ? 0 : kinds[a] > 0 ? hi_idx[a] : P.owned_lo[a];
      e[a] = kinds[a] < 0 ? 0 : hi_idx[a];
    }
    std::vector<std::int32_t> out;
    for (int K = b[2]; K <= e[2]; ++K)
      for (int J = b[1]; J <= e[1]; ++J)
        for (int I = b[0]; I <= e[0]; ++I) out.push_back((std::int32_t)(I + NX * (J + NY * K)));
    return out;
  };
  std::map<int, std::vector<std::int32_t>> send, recv;   // keyed by neighbour rank (ascending)
  for (int dx = 0; dx < 2; ++dx)
    for (int dy = 0; dy < 2; ++dy)
      for (int dz = 0; dz < 2; ++dz) {
        if (!dx && !dy && !dz) continue;
        const int d[3] = {dx, dy, dz}, md[3] = {-dx, -dy, -dz};
        std::array<int, 3> up, dn;
        bool up_ok = true, dn_ok = true;
        for (int a = 0; a < 3; ++a) {
          up[a] = P.coords[a] + d[a];
          dn[a] = P.coords[a] - d[a];
          if (periodic[a]) {
            up[a] = (up[a] + P.procs[a]) % P.procs[a];
            dn[a] = (dn[a] + P.procs[a]) % P.procs[a];
          }
          up_ok = up_ok && up[a] >= 0 && up[a] < P.procs[a];
          dn_ok = dn_ok && dn[a] >= 0 && dn[a] < P.procs[a];
        }
        // several directions can lead to the same neighbour (periodic): segments are
        // concatenated in direction order, identical on the sending and receiving side
        if (up_ok) {
          auto v = lattice_indices(d);
          auto& s = send[coords_rank(up, P.procs)];
          s.insert(s.end(), v.begin(), v.end());
        }
        if (dn_ok) {
          auto v = lattice_indices(md);
          auto& r = recv[coords_rank(dn, P.procs)];
          r.insert(r.end(), v.begin(), v.end());
        }
      }
  GhostLists& g = P.ghosts;
  g.ndofs = P.V.ndofs();
  for (auto& kv : send) {
    g.send_neighbors.push_back(kv.first);
    g.send_indices.insert(g.send_indices.end(), kv.second.begin(), kv.second.end());
    g.send_offsets.push_back((std::int32_t)g.send_indices.size());
  }
  for (auto& kv : recv) {
    g.recv_neighbors.push_back(kv.first);
    g.ghost_positions.insert(g.ghost_positions.end(), kv.second.begin(), kv.second.end());
    g.recv_offsets.push_back((std::int32_t)g.ghost_positions.size());
  }
  return part;
}

/// demo/cpu_planar3d/main.cpp:48-66: h = smallest cell diameter, dt = CFL*h/(c0*P^2)
/// rounded to an integer number of steps per source period.
inline std::pair<double, int> cfl_time_step(const BoxMesh& mesh, int degree, double c0, double freq, double CFL = 0.5)
{
  double hmin = 1e300;
  for (std::int32_t c = 0; c < mesh.ncells(); ++c) {
    double h = 0;
    for (int a = 0; a < 8; ++a)
      for (int b = a + 1; b < 8; ++b) {
        double d2 = 0;
        for (int d = 0; d < 3; ++d) {
          const double e = mesh.x[(std::size_t)mesh.geom_dofmap[(std::size_t)c * 8 + a] * 3 + d]
                           - mesh.x[(std::size_t)mesh.geom_dofmap[(std::size_t)c * 8 + b] * 3 + d];
          d2 += e * e;
        }
        h = std::max(h, std::sqrt(d2));
      }
    hmin = std::min(hmin, h);
  }
  double dt = CFL * hmin / (c0 * degree * degree);
  const double period = 1.0 / freq;
  const int stepPerPeriod = (int)(period / dt + 1);
  return {period / stepPerPeriod, stepPerPeriod};
}

}  // namespace wavehip
