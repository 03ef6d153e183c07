// Host-side work plan of the indexed marching kernels k_march_idx / k_march_idx_ks
// (stiffness_march_idx.hip, stiffness_march_ks.hip): the production kernels for an ARBITRARY
// dofmap.  StiffnessOperator(V, ...) of common/operators.hpp:137-201 visits its cells in dofmap
// order; cells are summed independently (operators.hpp:188-199), so the operator may regroup
// them freely -- and may look at every cell in whichever of the 48 orientations of the reference
// cube suits it, as long as dofs and geometry are relabelled together.
//
// The marching box kernel (stiffness_march.hip) owes its speed to columns of BX x BY x lz
// cells: the dof planes between the layers never leave the workgroup.  Nothing in that
// needs the lexicographic numbering -- only the ADDRESSES of the column's dofs do.  The
// plan therefore finds such columns in any conforming hexahedral mesh, whatever its cell order,
// dof numbering and local cell orientations (a DOLFINx box mesh, a gmsh multi-block mesh):
//   1. face links: two cells are linked when they share the four corner dofs of a face;
//   2. lattice placement: a breadth-first walk over the links gives every cell integer
//      coordinates (cx, cy, cz) and an ORIENTATION (signed permutation of its local axes) under
//      which its axes agree with the lattice.  A cell is only placed where every already placed
//      cell among its 26 lattice neighbours shares exactly the corner dofs the lattice says it
//      should (so irregular vertices -- three or five cells around an edge, O-grids -- and
//      periodic wrap-arounds cut the lattice into COMPONENTS instead of breaking the plan);
//   3. columns of BX x BY cells are cut into segments of <= lz layers = work items;
//   4. per item, the dof of every position of the column's dof tile
//      [P lz + 1][P BY + 1][P BX + 1] is recorded (as an offset from the item's smallest
//      dof, -1 where no cell covers the position) and checked for conformity: all cells
//      covering a position must name the same dof.  Identical tables are stored once
//      (PATTERNS), so a regularly numbered mesh keeps its index data in L2.
// The plan reports its fill (cells per cell slot); the caller falls back to the batch kernel
// k_stiffness_generic_u when the columns are mostly empty.
#include <algorithm>
#include <array>
#include <cstring>
#include <queue>
#include <unordered_map>

#include "common.h"

namespace wf {

// ---- cell orientations ---------------------------------------------------------------
// code = 8 * perm + flips.  Lattice axis m of the cell runs along the cell's own ("raw") axis
// kAxisPerm[perm][m], reversed when bit m of `flips` is set.
static const int kAxisPerm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};

void orient_decode(int code, int raw_axis[3], int flip[3])
{
  for (int m = 0; m < 3; ++m) {
    raw_axis[m] = kAxisPerm[code >> 3][m];
    flip[m] = (code >> m) & 1;
  }
}

// raw tensor index (x fastest) of the lattice-frame node (i, j, k), n nodes per direction
int orient_local_index(int code, int n, int i, int j, int k)
{
  int ra[3], fl[3];
  orient_decode(code, ra, fl);
  const int l[3] = {i, j, k};
  int r[3] = {0, 0, 0};
  for (int m = 0; m < 3; ++m) r[ra[m]] = fl[m] ? n - 1 - l[m] : l[m];
  return r[0] + n * (r[1] + n * r[2]);
}

// +1 for a rotation, -1 for a reflection of the reference cube
int orient_sign(int code)
{
  static const int perm_sign[6] = {1, -1, -1, 1, 1, -1};
  int s = perm_sign[code >> 3];
  for (int m = 0; m < 3; ++m)
    if ((code >> m) & 1) s = -s;
  return s;
}

namespace {

struct FaceRec {
  std::array<int32_t, 4> key;   // sorted corner dofs
  int32_t cell;
  int8_t face;                  // 2 * raw axis + side
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

// corner q (bits = lattice-frame position) of a cell under orientation `code`: index 0..7 of the raw corner
inline int oriented_corner(int code, int q)
{
  return orient_local_index(code, 2, q & 1, (q >> 1) & 1, (q >> 2) & 1);
}

}  // namespace

// tdm: tensor-ordered dofmap [ncells][nd] (the caller's cell frames).  Returns WF_OK and plan->ok =
// true when the mesh tiles into columns; plan->ok = false (still WF_OK) when it does not.
// normalise = false requires the cells' local axes to agree as given (orientation code 0 everywhere).
int build_march_plan(int P, size_t ncells, const int32_t* tdm, int BX, int BY, int lz_max, int lz_fixed, bool normalise,
                     MarchPlan* plan)
{
  int lz = lz_max;
  const int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY;
  const int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  plan->ok = false;
  plan->BX = BX;
  plan->BY = BY;
  plan->lz = lz_max;
  plan->nitems = 0;
  plan->reoriented = 0;
  plan->ncomponents = 0;
  plan->fill = 0.0;
  plan->cell_orient.assign(ncells, 0);
  if (ncells == 0) {
    plan->ok = true;
    plan->fill = 1.0;
    return WF_OK;
  }

  // corner dofs of every cell in its own frame, corner q = a + 2b + 4c
  std::vector<std::array<int32_t, 8>> corner(ncells);
  for (size_t c = 0; c < ncells; ++c) {
    const int32_t* d = tdm + c * nd;
    for (int q = 0; q < 8; ++q)
      corner[c][q] = d[((q & 1) ? P : 0) + n * (((q >> 1) & 1 ? P : 0) + n * ((q >> 2) & 1 ? P : 0))];
  }

  // ---- 1. face links: nb[c][raw face] = (cell, its raw face) ----------------------------
  std::vector<std::array<int32_t, 6>> nb(ncells);
  for (auto& a : nb) a.fill(-1);
  {
    std::vector<FaceRec> faces;
    faces.reserve(ncells * 6);
    for (size_t c = 0; c < ncells; ++c)
      for (int axis = 0; axis < 3; ++axis)
        for (int side = 0; side < 2; ++side) {
          FaceRec f;
          int m = 0;
          for (int q = 0; q < 8; ++q)
            if (((q >> axis) & 1) == side) f.key[m++] = corner[c][q];
          std::sort(f.key.begin(), f.key.end());
          f.cell = (int32_t)c;
          f.face = (int8_t)(2 * axis + side);
          faces.push_back(f);
        }
    std::sort(faces.begin(), faces.end());
    for (size_t a = 0; a + 1 < faces.size(); ++a) {
      const FaceRec &f = faces[a], &g = faces[a + 1];
      if (f.key != g.key || f.cell == g.cell) continue;
      if (a + 2 < faces.size() && faces[a + 2].key == f.key) return WF_OK;   // three cells on one face: not a manifold mesh
      if (f.key[0] == f.key[1] || f.key[1] == f.key[2] || f.key[2] == f.key[3]) continue;   // degenerate face
      nb[f.cell][f.face] = g.cell;
      nb[g.cell][g.face] = f.cell;
    }
  }
  std::vector<int32_t> key(ncells), by_key(ncells);
  for (size_t c = 0; c < ncells; ++c) {
    key[c] = *std::min_element(tdm + c * nd, tdm + (c + 1) * nd);
    by_key[c] = (int32_t)c;
  }
  std::stable_sort(by_key.begin(), by_key.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });

  // ---- 2. lattice placement (components, coordinates, orientations) -------------------
  std::vector<std::array<int32_t, 3>> xyz(ncells);
  std::vector<int32_t> comp(ncells, -1);
  std::vector<uint8_t>& orient = plan->cell_orient;
  std::vector<std::array<int32_t, 3>> cmin;
  std::unordered_map<std::array<int32_t, 4>, int32_t, KeyHash> at;   // (component, x, y, z) -> cell
  at.reserve(ncells * 2);

  // orientation of cell `o` such that its lattice-frame face (axis a, side 1 - s) carries, corner by
  // corner, the dofs `want[4]` (in-face positions (beta, gamma) over the other two lattice axes, beta fastest);
  // -1 if there is none
  auto orientation_from_face = [&](int32_t o, int a, int s, const int32_t want[4]) -> int {
    int r[4];
    for (int m = 0; m < 4; ++m) {
      r[m] = -1;
      for (int q = 0; q < 8; ++q)
        if (corner[o][q] == want[m]) {
          if (r[m] >= 0) return -1;   // a dof twice among the corners (periodic cell one cell wide)
          r[m] = q;
        }
      if (r[m] < 0) return -1;
    }
    const int b = (a + 1) % 3 < (a + 2) % 3 ? (a + 1) % 3 : (a + 2) % 3, c = 3 - a - b;   // in-face lattice axes, ascending
    const int db = r[1] ^ r[0], dc = r[2] ^ r[0];
    if ((db != 1 && db != 2 && db != 4) || (dc != 1 && dc != 2 && dc != 4) || db == dc || (r[3] ^ r[0]) != (db | dc)) return -1;
    const int rb = db == 1 ? 0 : db == 2 ? 1 : 2, rc = dc == 1 ? 0 : dc == 2 ? 1 : 2, ra = 3 - rb - rc;
    int raw_axis[3], flip[3];
    raw_axis[a] = ra;
    raw_axis[b] = rb;
    raw_axis[c] = rc;
    flip[b] = (r[0] >> rb) & 1;
    flip[c] = (r[0] >> rc) & 1;
    flip[a] = ((r[0] >> ra) & 1) ^ (1 - s);   // the shared face sits at lattice side 1 - s of the neighbour
    for (int pi = 0; pi < 6; ++pi)
      if (kAxisPerm[pi][0] == raw_axis[0] && kAxisPerm[pi][1] == raw_axis[1] && kAxisPerm[pi][2] == raw_axis[2])
        return 8 * pi + flip[0] + 2 * flip[1] + 4 * flip[2];
    return -1;
  };
  // lattice-frame corner dof of a placed cell
  auto cdof = [&](int32_t c, int q) { return corner[c][oriented_corner(orient[c], q)]; };
  // may cell c (with orientation oc) sit at p of component ci?  Every placed cell among the 26
  // lattice neighbours must share exactly the corner dofs the lattice prescribes.
  auto fits = [&](int32_t c, int oc, int32_t ci, const std::array<int32_t, 3>& p) {
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          if (!dx && !dy && !dz) continue;
          auto it = at.find({ci, p[0] + dx, p[1] + dy, p[2] + dz});
          if (it == at.end()) continue;
          const int32_t o = it->second;
          const int d[3] = {dx, dy, dz};
          for (int q = 0; q < 8; ++q) {   // corners of c that lie on the side(s) facing o
            bool shared = true;
            int qo = 0;
            for (int m = 0; m < 3; ++m) {
              const int bit = (q >> m) & 1;
              if (d[m] == 1 && !bit) shared = false;
              if (d[m] == -1 && bit) shared = false;
              qo |= (d[m] == 0 ? bit : 1 - bit) << m;
            }
            if (!shared) continue;
            if (corner[c][oriented_corner(oc, q)] != cdof(o, qo)) return false;
          }
        }
    return true;
  };

  for (size_t s0 = 0; s0 < ncells; ++s0) {
    const int32_t seed = by_key[s0];
    if (comp[seed] >= 0) continue;
    const int32_t ci = (int32_t)cmin.size();
    std::array<int32_t, 3> lo{0, 0, 0};
    std::queue<int32_t> q;
    comp[seed] = ci;
    xyz[seed] = {0, 0, 0};
    orient[seed] = 0;
    at[{ci, 0, 0, 0}] = seed;
    q.push(seed);
    while (!q.empty()) {
      const int32_t c = q.front();
      q.pop();
      for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], xyz[c][a]);
      int ra[3], fl[3];
      orient_decode(orient[c], ra, fl);
      for (int a = 0; a < 3; ++a)
        for (int s = 0; s < 2; ++s) {
          // lattice face (a, s) of c is its raw face (ra[a], s ^ fl[a])
          const int32_t o = nb[c][2 * ra[a] + (s ^ fl[a])];
          if (o < 0 || comp[o] >= 0) continue;   // placed cells were checked when they were placed
          std::array<int32_t, 3> want = xyz[c];
          want[a] += s ? 1 : -1;
          if (at.count({ci, want[0], want[1], want[2]})) continue;   // position taken: o starts another component later
          const int b = (a + 1) % 3 < (a + 2) % 3 ? (a + 1) % 3 : (a + 2) % 3, cc = 3 - a - b;
          int32_t fd[4];
          for (int m = 0; m < 4; ++m) fd[m] = cdof(c, (s << a) | ((m & 1) << b) | ((m >> 1) << cc));
          int oc = orientation_from_face(o, a, s, fd);
          if (oc < 0 || (!normalise && oc != 0)) continue;
          if (!fits(o, oc, ci, want)) continue;
          comp[o] = ci;
          xyz[o] = want;
          orient[o] = (uint8_t)oc;
          at[{ci, want[0], want[1], want[2]}] = o;
          q.push(o);
        }
    }
    cmin.push_back(lo);
  }
  plan->ncomponents = (int)cmin.size();
  for (size_t c = 0; c < ncells; ++c)
    if (orient[c] != 0) ++plan->reoriented;

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
  std::vector<std::array<int32_t, 4>> item_keys;   // (component, column x, column y, z segment) of items[]
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
      item_keys.push_back(ik);
    } else
      id = it->second;
    const int slot = (rz % lz) * CB + (ry % BY) * BX + (rx % BX);
    if (items[id][slot] >= 0) return WF_OK;   // two cells with the same coordinates (cannot happen: placement is injective)
    items[id][slot] = c;
  }
  const size_t nit = items.size();
  plan->fill = (double)ncells / ((double)nit * slots);
  std::vector<int32_t> item_min(nit, INT32_MAX), order(nit);
  for (size_t b = 0; b < nit; ++b) {
    for (int32_t c : items[b])
      if (c >= 0) item_min[b] = std::min(item_min[b], key[c]);
    order[b] = (int32_t)b;
  }
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return item_min[a] < item_min[b]; });

  // ---- 4. dof tiles, conformity check, patterns -----------------------------------
  // raw tensor index of the lattice-frame node (i, j, k) for each orientation in use
  std::vector<std::vector<int32_t>> lmap(48);
  auto local_map = [&](int code) -> const std::vector<int32_t>& {
    auto& m = lmap[code];
    if (m.empty()) {
      m.resize(nd);
      for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j)
          for (int i = 0; i < n; ++i) m[i + n * (j + n * k)] = orient_local_index(code, n, i, j, k);
    }
    return m;
  };
  const size_t tsize = (size_t)(P * lz + 1) * TP;
  plan->tile_size = (int)tsize;
  plan->slot_cell.assign(nit * slots, -1);
  plan->item_base.resize(nit);
  plan->item_pattern.resize(nit);
  plan->item_layers.resize(nit);
  plan->item_key.resize(nit);
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
          const std::vector<int32_t>& lm = local_map(orient[c]);
          for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i) {
                int32_t& slot = tile[(size_t)(P * l + k) * TP + (P * ly + j) * TX + (P * lx + i)];
                const int32_t dof = d[lm[i + n * (j + n * k)]];
                if (slot >= 0 && slot != dof) return WF_OK;   // the cells of the column do not conform on the tile
                slot = dof;
              }
        }
    // (a dof may sit at two positions of the tile -- around an irregular edge, or on a periodic mesh
    // one column wide: its x value is read twice and its contributions reach y in two atomics)
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
    plan->item_key[ob] = item_keys[order[ob]];
  }
  plan->nitems = (int)nit;
  plan->npatterns = npat;
  plan->ok = true;
  return WF_OK;
}

}  // namespace wf

// SURVEY section 7's "setup-time renumbering option": a dof numbering that follows the lattice plan -- work items in
// (component, z segment, column y, column x) order, inside an item plane by plane, row by row -- so that the
// rows the marching kernels read and add to are contiguous in memory whatever numbering the caller's space came
// with (a uniformly random numbering costs 2.8x in the stiffness apply: every access its own cache line).
// h_dofmap: tensor-ordered (x fastest) [ncells][(P+1)^3]; h_new_of_old[ndofs]: new index of every old dof
// (dofs no cell refers to keep their relative order behind the others).  A mesh that does not tile into lattice
// columns gets the first-touch numbering of its cell order.
extern "C" int wf_lattice_numbering(int degree, int64_t ncells, int32_t ndofs, const int32_t* h_dofmap, int32_t* h_new_of_old)
{
  using namespace wf;
  WF_REQUIRE(degree >= 1 && degree <= 7, "wf_lattice_numbering: degree must be 1..7");
  WF_REQUIRE(ncells >= 0 && ndofs >= 0 && (ncells == 0 || h_dofmap) && (ndofs == 0 || h_new_of_old), "wf_lattice_numbering: bad arguments");
  const int P = degree, n = P + 1, nd = n * n * n;
  for (int64_t e = 0; e < ncells * nd; ++e)
    WF_REQUIRE(h_dofmap[e] >= 0 && h_dofmap[e] < ndofs, "wf_lattice_numbering: dofmap entry out of range");
  std::fill(h_new_of_old, h_new_of_old + ndofs, -1);
  int32_t next = 0;
  auto touch = [&](int32_t d) {
    if (h_new_of_old[d] < 0) h_new_of_old[d] = next++;
  };
  int BX = 0, BY = 0;
  march_idx_shape(OP_KIND_STIFFNESS, P, &BX, &BY);
  MarchPlan plan;
  int rc = build_march_plan(P, (size_t)ncells, h_dofmap, BX, BY, 8, 8, true, &plan);
  if (rc != WF_OK) return rc;
  if (plan.ok && ncells > 0) {
    std::vector<int32_t> order(plan.nitems);
    for (int i = 0; i < plan.nitems; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
      const auto &ka = plan.item_key[a], &kb = plan.item_key[b];
      if (ka[0] != kb[0]) return ka[0] < kb[0];
      if (ka[3] != kb[3]) return ka[3] < kb[3];
      if (ka[2] != kb[2]) return ka[2] < kb[2];
      return ka[1] < kb[1];
    });
    for (int32_t it : order) {
      const int32_t* pat = &plan.pat_off[(size_t)plan.item_pattern[it] * plan.tile_size];
      for (int e = 0; e < plan.tile_size; ++e)
        if (pat[e] >= 0) touch(plan.item_base[it] + pat[e]);
    }
  } else {
    for (int64_t e = 0; e < ncells * nd; ++e) touch(h_dofmap[e]);
  }
  for (int32_t d = 0; d < ndofs; ++d) touch(d);
  return WF_OK;
}
