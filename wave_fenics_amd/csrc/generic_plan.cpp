// Host-side work plan of the indexed marching stiffness kernel k_stiffness_march_idx
// (stiffness_march_idx.hip): the production kernel for an ARBITRARY dofmap.
// StiffnessOperator(V, ...) of common/operators.hpp:137-201 visits its cells in dofmap
// order; cells are summed independently (operators.hpp:188-199), so the operator may
// regroup them freely.
//
// The marching box kernel (stiffness_march.hip) owes its speed to columns of BX x BY x lz
// cells: the dof planes between the layers never leave the workgroup.  Nothing in that
// needs the lexicographic numbering -- only the ADDRESSES of the column's dofs do.  The
// plan therefore finds such columns in any conforming hexahedral mesh whose cells link up
// like a lattice, whatever its cell order and dof numbering (e.g. a DOLFINx box mesh):
//   1. lattice detection: cells are linked through faces whose four corner dofs and
//      orientation agree; a breadth-first walk over the links gives every cell integer
//      coordinates (cx, cy, cz);
//   2. columns of BX x BY cells are cut into segments of <= lz layers = work items;
//   3. per item, the dof of every position of the column's dof tile
//      [P lz + 1][P BY + 1][P BX + 1] is recorded (as an offset from the item's smallest
//      dof, -1 where no cell covers the position) and checked for conformity: all cells
//      covering a position must name the same dof.  Identical tables are stored once
//      (PATTERNS), so a regularly numbered mesh keeps its index data in L2.
// Meshes that do not link up (or do not tile consistently) are reported as such; the
// caller falls back to the batch kernel k_stiffness_generic_u.
#include <algorithm>
#include <array>
#include <cstring>
#include <queue>
#include <unordered_map>

#include "common.h"

namespace wf {

namespace {

struct FaceRec {
  std::array<int32_t, 4> key;   // sorted corner dofs
  std::array<int32_t, 4> ord;   // corner dofs in the cell's tensor order
  int32_t cell;
  int8_t axis, side;
  bool operator<(const FaceRec& o) const { return key < o.key; }
};

uint64_t fnv(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull)
{
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < bytes; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}

struct KeyHash {
  size_t operator()(const std::array<int32_t, 4>& k) const { return (size_t)fnv(k.data(), sizeof(k)); }
};

}  // namespace

// tdm: tensor-ordered dofmap [ncells][nd].  Returns WF_OK and plan->ok = true when the mesh
// tiles into columns; plan->ok = false (still WF_OK) when it does not.
int build_march_plan(int P, size_t ncells, const int32_t* tdm, int BX, int BY, int lz_max, int lz_fixed, MarchPlan* plan)
{
  int lz = lz_max;
  const int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY;
  const int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  plan->ok = false;
  plan->BX = BX;
  plan->BY = BY;
  plan->lz = lz_max;
  plan->nitems = 0;
  if (ncells == 0) {
    plan->ok = true;
    return WF_OK;
  }

  // ---- 1. face links -----------------------------------------------------------
  std::vector<std::array<int32_t, 6>> nb(ncells);
  for (auto& a : nb) a.fill(-1);
  {
    std::vector<FaceRec> faces;
    faces.reserve(ncells * 6);
    for (size_t c = 0; c < ncells; ++c) {
      const int32_t* d = tdm + c * nd;
      int32_t v[8];
      for (int q = 0; q < 8; ++q) v[q] = d[((q & 1) ? P : 0) + n * (((q >> 1) & 1 ? P : 0) + n * ((q >> 2) & 1 ? P : 0))];
      for (int axis = 0; axis < 3; ++axis)
        for (int side = 0; side < 2; ++side) {
          FaceRec f;
          int m = 0;
          for (int q = 0; q < 8; ++q)
            if (((q >> axis) & 1) == side) f.ord[m++] = v[q];
          f.key = f.ord;
          std::sort(f.key.begin(), f.key.end());
          f.cell = (int32_t)c;
          f.axis = (int8_t)axis;
          f.side = (int8_t)side;
          faces.push_back(f);
        }
    }
    std::sort(faces.begin(), faces.end());
    for (size_t a = 0; a + 1 < faces.size(); ++a) {
      const FaceRec &f = faces[a], &g = faces[a + 1];
      if (f.key != g.key) continue;
      if (a + 2 < faces.size() && faces[a + 2].key == f.key) return WF_OK;   // non-manifold: not a lattice
      // consistent orientation: same axis, opposite sides, same in-face corner order
      if (f.axis == g.axis && f.side != g.side && f.ord == g.ord) {
        nb[f.cell][2 * f.axis + f.side] = g.cell;
        nb[g.cell][2 * g.axis + g.side] = f.cell;
      } else if (f.cell != g.cell) {
        return WF_OK;   // two cells meet with different orientations: no global lattice
      }
    }
  }
  std::vector<int32_t> key(ncells), by_key(ncells);
  for (size_t c = 0; c < ncells; ++c) {
    key[c] = *std::min_element(tdm + c * nd, tdm + (c + 1) * nd);
    by_key[c] = (int32_t)c;
  }
  std::stable_sort(by_key.begin(), by_key.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });

  // ---- 2. lattice coordinates (per connected component) -----------------------------
  std::vector<std::array<int32_t, 3>> xyz(ncells);
  std::vector<int32_t> comp(ncells, -1);
  std::vector<std::array<int32_t, 3>> cmin;
  for (size_t s0 = 0; s0 < ncells; ++s0) {
    const int32_t seed = by_key[s0];
    if (comp[seed] >= 0) continue;
    const int32_t ci = (int32_t)cmin.size();
    std::array<int32_t, 3> lo{0, 0, 0};
    std::queue<int32_t> q;
    comp[seed] = ci;
    xyz[seed] = {0, 0, 0};
    q.push(seed);
    while (!q.empty()) {
      const int32_t c = q.front();
      q.pop();
      for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], xyz[c][a]);
      for (int f = 0; f < 6; ++f) {
        const int32_t o = nb[c][f];
        if (o < 0) continue;
        std::array<int32_t, 3> want = xyz[c];
        want[f / 2] += (f & 1) ? 1 : -1;
        if (comp[o] >= 0) {
          if (xyz[o] != want) return WF_OK;   // the links close a loop that is not a lattice loop (e.g. periodic)
          continue;
        }
        comp[o] = ci;
        xyz[o] = want;
        q.push(o);
      }
    }
    cmin.push_back(lo);
  }

  // ---- 3. columns and z segments = work items ----------------------------------
  // Segment length: work items run in rounds of the 512 resident workgroups (2 per CU) and each
  // pays ~1.5 layers of pipeline fill; pick the lz <= lz_max that minimises rounds * (lz + 1.5)
  // (the rule of the box operator, api.hip).
  if (lz_fixed > 0) {
    lz = std::min(lz_max, lz_fixed);
  } else {
    std::unordered_map<std::array<int32_t, 4>, std::array<int32_t, 2>, KeyHash> cols;   // column -> z range
    for (size_t c = 0; c < ncells; ++c) {
      const auto& lo = cmin[comp[c]];
      const std::array<int32_t, 4> ck{comp[c], (xyz[c][0] - lo[0]) / BX, (xyz[c][1] - lo[1]) / BY, 0};
      const int rz = xyz[c][2] - lo[2];
      auto it = cols.find(ck);
      if (it == cols.end())
        cols.emplace(ck, std::array<int32_t, 2>{rz, rz});
      else {
        it->second[0] = std::min(it->second[0], rz);
        it->second[1] = std::max(it->second[1], rz);
      }
    }
    double best = 1e300;
    for (int cand = 1; cand <= lz_max; ++cand) {
      long nitems = 0;
      for (const auto& kv : cols) nitems += kv.second[1] / cand - kv.second[0] / cand + 1;
      const double cost = (double)((nitems + 511) / 512) * (cand + 1.5);
      if (cost <= best + 1e-9) {
        best = cost;
        lz = cand;
      }
    }
  }
  plan->lz = lz;
  std::unordered_map<std::array<int32_t, 4>, int32_t, KeyHash> item_of;
  std::vector<std::vector<int32_t>> items;   // cells in slot order [layer][ly][lx], -1 = missing
  const int slots = lz * CB;
  for (size_t s0 = 0; s0 < ncells; ++s0) {
    const int32_t c = by_key[s0];
    const auto& lo = cmin[comp[c]];
    const int rx = xyz[c][0] - lo[0], ry = xyz[c][1] - lo[1], rz = xyz[c][2] - lo[2];
    const std::array<int32_t, 4> ik{comp[c], rx / BX, ry / BY, rz / lz};
    auto it = item_of.find(ik);
    int32_t id;
    if (it == item_of.end()) {
      id = (int32_t)items.size();
      item_of.emplace(ik, id);
      items.emplace_back(slots, -1);
    } else
      id = it->second;
    const int slot = (rz % lz) * CB + (ry % BY) * BX + (rx % BX);
    if (items[id][slot] >= 0) return WF_OK;   // two cells with the same coordinates
    items[id][slot] = c;
  }
  const size_t nit = items.size();
  std::vector<int32_t> item_min(nit, INT32_MAX), order(nit);
  for (size_t b = 0; b < nit; ++b) {
    for (int32_t c : items[b])
      if (c >= 0) item_min[b] = std::min(item_min[b], key[c]);
    order[b] = (int32_t)b;
  }
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return item_min[a] < item_min[b]; });

  // ---- 4. dof tiles, conformity check, patterns -----------------------------------
  const size_t tsize = (size_t)(P * lz + 1) * TP;
  plan->tile_size = (int)tsize;
  plan->slot_cell.assign(nit * slots, -1);
  plan->item_base.resize(nit);
  plan->item_pattern.resize(nit);
  plan->item_layers.resize(nit);
  plan->pat_off.clear();
  std::unordered_map<uint64_t, std::vector<int32_t>> seen;
  std::vector<int32_t> tile(tsize);
  int npat = 0;
  for (size_t ob = 0; ob < nit; ++ob) {
    const std::vector<int32_t>& cells = items[order[ob]];
    std::copy(cells.begin(), cells.end(), plan->slot_cell.begin() + ob * slots);
    std::fill(tile.begin(), tile.end(), -1);
    int layers = 0;
    for (int l = 0; l < lz; ++l)
      for (int ly = 0; ly < BY; ++ly)
        for (int lx = 0; lx < BX; ++lx) {
          const int32_t c = cells[l * CB + ly * BX + lx];
          if (c < 0) continue;
          layers = l + 1;
          const int32_t* d = tdm + (size_t)c * nd;
          for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i) {
                int32_t& slot = tile[(size_t)(P * l + k) * TP + (P * ly + j) * TX + (P * lx + i)];
                const int32_t dof = d[i + n * (j + n * k)];
                if (slot >= 0 && slot != dof) return WF_OK;   // the cells of the column do not conform on the tile
                slot = dof;
              }
        }
    // a dof must not sit at two positions of the tile (its contributions would be added twice in
    // different flushes -- harmless -- but its x value is simply read twice; allowed)
    const int32_t base = item_min[order[ob]];
    for (auto& v : tile)
      if (v >= 0) v -= base;
    const uint64_t h = fnv(tile.data(), tsize * sizeof(int32_t));
    int32_t pid = -1;
    auto& cand = seen[h];
    for (int32_t p : cand)
      if (std::memcmp(&plan->pat_off[(size_t)p * tsize], tile.data(), tsize * sizeof(int32_t)) == 0) {
        pid = p;
        break;
      }
    if (pid < 0) {
      pid = npat++;
      cand.push_back(pid);
      plan->pat_off.insert(plan->pat_off.end(), tile.begin(), tile.end());
    }
    plan->item_base[ob] = base;
    plan->item_pattern[ob] = pid;
    plan->item_layers[ob] = layers;
  }
  plan->nitems = (int)nit;
  plan->npatterns = npat;
  plan->ok = true;
  return WF_OK;
}

}  // namespace wf
