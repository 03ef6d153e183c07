// Host side of the mesh-file path: what the reference's driver gets from DOLFINx after
// reading mesh + facet tags (demo/cpu_planar3d/main.cpp:39-66, common/LinearGLL.hpp:113-115) --
// the degree-P function space of an arbitrary conforming hexahedral mesh, the tagged boundary
// facets, their collocated facet masses (diagonal GLL form of forms.ufl:19-24) and mesh::h.
// Host-only C ABI (wf_fs_*), used by the Python package (mesh_io.py) and by the C++ mirror
// (include/wavehip_mesh.hpp).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <unordered_map>

#include "common.h"

using namespace wf;

namespace {

struct Key4Hash {
  size_t operator()(const std::array<int32_t, 4>& k) const
  {
    uint64_t h = 1469598103934665603ull;
    for (int32_t v : k) {
      h ^= (uint32_t)v;
      h *= 1099511628211ull;
    }
    return (size_t)h;
  }
};

// trilinear map of a cell: x(X) = sum_v N_v(X) x_v, vertex v = a + 2b + 4c
inline void q1_point(const double (*xv)[3], double X0, double X1, double X2, double* out)
{
  out[0] = out[1] = out[2] = 0.0;
  for (int v = 0; v < 8; ++v) {
    const double N = ((v & 1) ? X0 : 1.0 - X0) * ((v & 2) ? X1 : 1.0 - X1) * ((v & 4) ? X2 : 1.0 - X2);
    for (int d = 0; d < 3; ++d) out[d] += N * xv[v][d];
  }
}

// d x / d X_d at X
inline void q1_tangent(const double (*xv)[3], const double X[3], int d, double* out)
{
  out[0] = out[1] = out[2] = 0.0;
  for (int v = 0; v < 8; ++v) {
    double f = 1.0;
    for (int dd = 0; dd < 3; ++dd) {
      const bool up = (v >> dd) & 1;
      f *= dd == d ? (up ? 1.0 : -1.0) : (up ? X[dd] : 1.0 - X[dd]);
    }
    for (int e = 0; e < 3; ++e) out[e] += f * xv[v][e];
  }
}

const int kFaceVerts[6][4] = {{0, 2, 4, 6}, {1, 3, 5, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {0, 1, 2, 3}, {4, 5, 6, 7}};

}  // namespace

extern "C" {

// fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped)) for an arbitrary
// conforming hexahedral mesh with cells in ANY local orientation.  Dofs are identified
// TOPOLOGICALLY -- a vertex dof by its vertex, an edge dof by the edge's two vertices and its
// position counted from the smaller vertex id, a face dof by the face's vertices and its position
// in the frame spanned from the face's smallest vertex towards its smaller neighbour, interior
// dofs by the cell -- so two cells that see a shared entity in different vertex orders agree
// exactly (coordinates computed through different trilinear sums differ in the last bits and
// must not decide identity).  The dofs are then numbered in lexicographic order of their
// coordinates (z, y, x), which only decides the ORDER.  h_dofmap [ncells][(P+1)^3] in tensor order
// (x fastest) of each cell's own frame; h_dof_coords [capacity >= *ndofs][3] may be NULL.
int wf_fs_build(int degree, int64_t nverts, const double* h_xverts, int64_t ncells, const int32_t* h_cells,
                int64_t* ndofs, int32_t* h_dofmap, double* h_dof_coords, int64_t coords_capacity)
{
  WF_REQUIRE(degree >= 1 && degree <= kMaxDegree, "wf_fs_build: degree must be 1..7");
  WF_REQUIRE(nverts >= 0 && ncells >= 0 && ndofs && (ncells == 0 || (h_xverts && h_cells && h_dofmap)), "wf_fs_build: bad argument");
  const int P = degree, n = P + 1, nd = n * n * n, m = P - 1;
  for (int64_t e = 0; e < ncells * 8; ++e)
    WF_REQUIRE(h_cells[e] >= 0 && h_cells[e] < nverts, "wf_fs_build: vertex index out of range");
  std::vector<double> pts(n), wts(n);
  gll_points_weights(n, pts.data(), wts.data());

  // provisional topological ids: [vertices | edges x m | faces x m^2 | cells x m^3]
  std::unordered_map<std::array<int32_t, 4>, int32_t, Key4Hash> edge_id, face_id;
  edge_id.reserve((size_t)ncells * 4);
  face_id.reserve((size_t)ncells * 4);
  std::vector<int64_t> tid((size_t)ncells * nd);
  auto tensor = [&](int i, int j, int k) { return i + n * (j + n * k); };
  // pass 1: register edges and faces
  for (int64_t c = 0; c < ncells; ++c) {
    const int32_t* v = h_cells + c * 8;
    for (int d = 0; d < 3; ++d)                  // 4 edges along axis d
      for (int q = 0; q < 8; ++q) {
        if ((q >> d) & 1) continue;
        const int32_t a = v[q], b = v[q | (1 << d)];
        edge_id.emplace(std::array<int32_t, 4>{std::min(a, b), std::max(a, b), -1, -1}, (int32_t)edge_id.size());
      }
    for (int f = 0; f < 6; ++f) {
      std::array<int32_t, 4> key{v[kFaceVerts[f][0]], v[kFaceVerts[f][1]], v[kFaceVerts[f][2]], v[kFaceVerts[f][3]]};
      std::sort(key.begin(), key.end());
      face_id.emplace(key, (int32_t)face_id.size());
    }
  }
  const int64_t base_e = nverts, base_f = base_e + (int64_t)edge_id.size() * m;
  const int64_t base_c = base_f + (int64_t)face_id.size() * m * m, total = base_c + ncells * (int64_t)m * m * m;
  WF_REQUIRE(total < ((int64_t)1 << 31), "wf_fs_build: more than 2^31 dofs");
  // pass 2: ids of every element-local dof
  for (int64_t c = 0; c < ncells; ++c) {
    const int32_t* v = h_cells + c * 8;
    int64_t* t = &tid[(size_t)c * nd];
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          const int l[3] = {i, j, k};
          int nint = 0, iax[3] = {0, 0, 0};
          int corner = 0;   // corner bits of the boundary coordinates
          for (int d = 0; d < 3; ++d) {
            if (l[d] > 0 && l[d] < P)
              iax[nint++] = d;
            else if (l[d] == P)
              corner |= 1 << d;
          }
          int64_t id;
          if (nint == 0) {
            id = v[corner];
          } else if (nint == 1) {
            const int d = iax[0];
            const int32_t a = v[corner], b = v[corner | (1 << d)];
            const int pos = a < b ? l[d] : P - l[d];   // counted from the smaller vertex (the GLL points are symmetric)
            id = base_e + (int64_t)edge_id.at({std::min(a, b), std::max(a, b), -1, -1}) * m + (pos - 1);
          } else if (nint == 2) {
            const int da = iax[0], db = iax[1];
            const int32_t g[4] = {v[corner], v[corner | (1 << da)], v[corner | (1 << db)], v[corner | (1 << da) | (1 << db)]};
            std::array<int32_t, 4> key{g[0], g[1], g[2], g[3]};
            std::sort(key.begin(), key.end());
            int mn = 0;
            for (int q = 1; q < 4; ++q)
              if (g[q] < g[mn]) mn = q;
            const int am = mn & 1, bm = mn >> 1;
            const int32_t na = g[(1 - am) | (bm << 1)], nb = g[am | ((1 - bm) << 1)];
            const int s = am ? P - l[da] : l[da], t2 = bm ? P - l[db] : l[db];
            const int u = na < nb ? s : t2, w = na < nb ? t2 : s;
            id = base_f + (int64_t)face_id.at(key) * m * m + (u - 1) + (int64_t)m * (w - 1);
          } else {
            id = base_c + c * (int64_t)m * m * m + (i - 1) + (int64_t)m * ((j - 1) + (int64_t)m * (k - 1));
          }
          t[tensor(i, j, k)] = id;
        }
  }
  // compact to the ids in use (vertices not referenced by any cell drop out) and take each dof's coordinates
  // from its first occurrence
  std::vector<int32_t> first((size_t)total, -1);
  std::vector<double> X;
  std::vector<int64_t> used;
  used.reserve((size_t)total);
  for (int64_t c = 0; c < ncells; ++c) {
    double xv[8][3];
    for (int q = 0; q < 8; ++q)
      for (int d = 0; d < 3; ++d) xv[q][d] = h_xverts[(size_t)h_cells[c * 8 + q] * 3 + d];
    for (int l = 0; l < nd; ++l) {
      const int64_t id = tid[(size_t)c * nd + l];
      if (first[id] >= 0) continue;
      first[id] = (int32_t)used.size();
      used.push_back(id);
      double p3[3];
      q1_point(xv, pts[l % n], pts[(l / n) % n], pts[l / (n * n)], p3);
      X.insert(X.end(), p3, p3 + 3);
    }
  }
  const int64_t N = (int64_t)used.size();
  // order: lexicographic (z, y, x) of coordinates quantised to 1e-9 of the shortest cell edge, ties by id
  double emin = 1e300;
  for (int64_t c = 0; c < ncells; ++c) {
    const int32_t* v = h_cells + c * 8;
    for (int d = 0; d < 3; ++d) {
      double s = 0.0;
      for (int e = 0; e < 3; ++e) {
        const double dx = h_xverts[(size_t)v[1 << d] * 3 + e] - h_xverts[(size_t)v[0] * 3 + e];
        s += dx * dx;
      }
      emin = std::min(emin, std::sqrt(s));
    }
  }
  const double q = ncells ? 1.0 / (1e-9 * emin) : 1.0;
  std::vector<std::array<int64_t, 3>> qx((size_t)N);
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < 3; ++d) qx[i][d] = (int64_t)std::llround(X[(size_t)i * 3 + d] * q);
  std::vector<int32_t> order((size_t)N), rank((size_t)N);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    if (qx[a][2] != qx[b][2]) return qx[a][2] < qx[b][2];
    if (qx[a][1] != qx[b][1]) return qx[a][1] < qx[b][1];
    return qx[a][0] < qx[b][0];
  });
  for (int64_t r = 0; r < N; ++r) rank[order[r]] = (int32_t)r;
  for (size_t e = 0; e < (size_t)ncells * nd; ++e) h_dofmap[e] = rank[first[tid[e]]];
  *ndofs = N;
  if (h_dof_coords) {
    WF_REQUIRE(coords_capacity >= N, "wf_fs_build: dof coordinate buffer too small");
    for (int64_t i = 0; i < N; ++i)
      for (int d = 0; d < 3; ++d) h_dof_coords[(size_t)rank[i] * 3 + d] = X[(size_t)i * 3 + d];
  }
  return WF_OK;
}

// The (cell, axis, side) of every facet given by its four vertices (any order): the facet's vertex
// set is matched against the faces of the cells (an exterior facet belongs to one cell; of two
// cells the first is reported).
int wf_fs_locate_facets(int64_t ncells, const int32_t* h_cells, int64_t nfacets, const int32_t* h_facet_verts,
                        int32_t* h_cell, int32_t* h_axis, int32_t* h_side)
{
  WF_REQUIRE(ncells >= 0 && nfacets >= 0 && (nfacets == 0 || (h_cells && h_facet_verts && h_cell && h_axis && h_side)),
             "wf_fs_locate_facets: bad argument");
  std::unordered_map<std::array<int32_t, 4>, std::array<int32_t, 2>, Key4Hash> face_of;
  face_of.reserve((size_t)ncells * 6);
  for (int64_t c = 0; c < ncells; ++c)
    for (int f = 0; f < 6; ++f) {
      std::array<int32_t, 4> key;
      for (int q = 0; q < 4; ++q) key[q] = h_cells[c * 8 + kFaceVerts[f][q]];
      std::sort(key.begin(), key.end());
      face_of.emplace(key, std::array<int32_t, 2>{(int32_t)c, f});
    }
  for (int64_t i = 0; i < nfacets; ++i) {
    std::array<int32_t, 4> key{h_facet_verts[i * 4], h_facet_verts[i * 4 + 1], h_facet_verts[i * 4 + 2], h_facet_verts[i * 4 + 3]};
    std::sort(key.begin(), key.end());
    auto it = face_of.find(key);
    if (it == face_of.end()) {
      set_error("wf_fs_locate_facets: a tagged facet is not a face of any cell");
      return WF_ERR_INVALID;
    }
    h_cell[i] = it->second[0];
    h_axis[i] = it->second[1] / 2;
    h_side[i] = it->second[1] % 2;
  }
  return WF_OK;
}

// Collocated facet masses m[i] = sum_facets w_q |dx/ds x dx/dt| (diagonal GLL form of
// inner(g, v) * ds(tag), demo/cpu_planar3d/forms.ufl:19-24) of a list of facets (cell, axis, side).
// Output: the dofs touched, ascending, and their masses; capacity nfacets (P+1)^2 is always enough.
int wf_fs_facet_mass(int degree, int64_t nverts, const double* h_xverts, int64_t ncells, const int32_t* h_cells,
                     const int32_t* h_dofmap, int64_t nfacets, const int32_t* h_cell, const int32_t* h_axis,
                     const int32_t* h_side, int64_t* nout, int32_t* h_idx, double* h_mass)
{
  WF_REQUIRE(degree >= 1 && degree <= kMaxDegree && nout, "wf_fs_facet_mass: bad argument");
  WF_REQUIRE(nfacets == 0 || (h_xverts && h_cells && h_dofmap && h_cell && h_axis && h_side && h_idx && h_mass),
             "wf_fs_facet_mass: null array");
  const int P = degree, n = P + 1, nd = n * n * n;
  std::vector<double> pts(n), wts(n);
  gll_points_weights(n, pts.data(), wts.data());
  std::map<int32_t, double> acc;
  for (int64_t f = 0; f < nfacets; ++f) {
    const int64_t c = h_cell[f];
    const int axis = h_axis[f], side = h_side[f];
    WF_REQUIRE(c >= 0 && c < ncells && axis >= 0 && axis < 3 && (side == 0 || side == 1), "wf_fs_facet_mass: bad facet");
    const int ta = axis == 0 ? 1 : 0, tb = axis == 2 ? 1 : 2;
    double xv[8][3];
    for (int q = 0; q < 8; ++q) {
      const int32_t v = h_cells[c * 8 + q];
      WF_REQUIRE(v >= 0 && v < nverts, "wf_fs_facet_mass: vertex index out of range");
      for (int d = 0; d < 3; ++d) xv[q][d] = h_xverts[(size_t)v * 3 + d];
    }
    for (int b = 0; b < n; ++b)
      for (int a = 0; a < n; ++a) {
        double X[3];
        X[axis] = (double)side;
        X[ta] = pts[a];
        X[tb] = pts[b];
        double t0[3], t1[3];
        q1_tangent(xv, X, ta, t0);
        q1_tangent(xv, X, tb, t1);
        const double cx = t0[1] * t1[2] - t0[2] * t1[1], cy = t0[2] * t1[0] - t0[0] * t1[2], cz = t0[0] * t1[1] - t0[1] * t1[0];
        const double ds = std::sqrt(cx * cx + cy * cy + cz * cz) * wts[a] * wts[b];
        int l[3];
        l[axis] = side * P;
        l[ta] = a;
        l[tb] = b;
        acc[h_dofmap[c * nd + l[0] + n * (l[1] + n * l[2])]] += ds;
      }
  }
  int64_t k = 0;
  for (const auto& kv : acc) {
    h_idx[k] = kv.first;
    h_mass[k] = kv.second;
    ++k;
  }
  *nout = k;
  return WF_OK;
}

// mesh::h of demo/cpu_planar3d/main.cpp:48-57: per cell the largest distance between two of its
// vertices; returns the minimum over the cells.
int wf_fs_min_cell_diameter(int64_t nverts, const double* h_xverts, int64_t ncells, const int32_t* h_cells, double* hmin)
{
  WF_REQUIRE(hmin && (ncells == 0 || (h_xverts && h_cells)), "wf_fs_min_cell_diameter: bad argument");
  double best = 1e300;
  for (int64_t c = 0; c < ncells; ++c) {
    double dmax = 0.0;
    for (int a = 0; a < 8; ++a)
      for (int b = a + 1; b < 8; ++b) {
        const int32_t va = h_cells[c * 8 + a], vb = h_cells[c * 8 + b];
        WF_REQUIRE(va >= 0 && va < nverts && vb >= 0 && vb < nverts, "wf_fs_min_cell_diameter: vertex index out of range");
        double s = 0.0;
        for (int d = 0; d < 3; ++d) {
          const double dx = h_xverts[(size_t)va * 3 + d] - h_xverts[(size_t)vb * 3 + d];
          s += dx * dx;
        }
        dmax = std::max(dmax, s);
      }
    best = std::min(best, std::sqrt(dmax));
  }
  *hmin = best;
  return WF_OK;
}

}  // extern "C"
