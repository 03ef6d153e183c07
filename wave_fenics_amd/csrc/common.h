// Internal declarations shared by the libwavehip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "wavehip.h"

// Diagnostic ablation flags (WF_ABLATE environment variable: skip the scatter, serve geometry from
// L2, ...) are compiled in only with -DWF_DIAG (tools/diag_build.sh).  In the product build the
// kernels see the constant 0 and every diagnostic branch folds away: as run-time branches they
// split the hot loops into basic blocks, and at each join the compiler's s_waitcnt bookkeeping
// turns conservative (measured +13 % on the tetrahedral MFMA kernel).
#ifdef WF_DIAG
#define WF_ABLATE_FLAGS(arg) (arg)
#else
#define WF_ABLATE_FLAGS(arg) 0
#endif

namespace wf {

constexpr int kMaxDegree = 7;
constexpr int kMaxN = kMaxDegree + 1;

void set_error(const std::string& msg);

// roctx range markers (markers.cpp); no-ops unless wf_markers_enable(1)
bool markers_on();
void marker_push(const char* name);
void marker_pop();
struct MarkerScope {
  explicit MarkerScope(const char* name) { marker_push(name); }
  ~MarkerScope() { marker_pop(); }
};

#define WF_HIP_CHECK(expr)                                                          \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ::wf::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));     \
      return WF_ERR_HIP;                                                            \
    }                                                                               \
  } while (0)

#define WF_REQUIRE(cond, msg)                    \
  do {                                           \
    if (!(cond)) {                               \
      ::wf::set_error(msg);                      \
      return WF_ERR_INVALID;                     \
    }                                            \
  } while (0)

// 1-D collocation derivative matrix, passed by value as a kernel argument so
// that compile-time-indexed entries become scalar (SGPR) operands.
struct DMat {
  double v[kMaxN * kMaxN];
};

// ---- host tabulation (tables.cpp) ----
void gll_points_weights(int n, double* pts, double* wts);
void gll_derivative_matrix(int P, double* D);  // clamped, D[q*n + a]

// cells per workgroup batch of the column-thread kernels: floor(256 / n^2)
inline int cells_per_batch(int P) { return 256 / ((P + 1) * (P + 1)); }

// ---- kernel launchers (kernels.hip) ----
int launch_geometry_hex(int P, int ncells, const double* d_xverts, const int32_t* d_geom_dofmap,
                        const double* d_pts, const double* d_wts, int use_fabs, int clamp,
                        double* d_G9, double* d_G6blk, double* d_detJ, hipStream_t s);
// lattice plans with a fill (cells per cell slot) below this keep the batch kernel: the marching kernels
// read geometry for every slot of a column, empty or not
constexpr double kMinPlanFill = 0.55;
int launch_geometry_box(int P, int nx, int ny, int nz, int bx, int by, int bz, const double* d_xverts,
                        const double* d_pts, const double* d_wts, int use_fabs, int clamp,
                        double* d_G6blk, double* d_detJ_lattice, hipStream_t s);
int launch_geometry_hex_slots(int P, int CB, int ncells, const double* d_xverts, const int32_t* d_geom_dofmap,
                              const int32_t* d_slot_of, const uint8_t* d_orient, const double* d_pts, const double* d_wts,
                              int use_fabs, int clamp, double* d_G6blk, hipStream_t s);
int launch_pack_G6(int P, int CB, int ncells, const double* d_G9, double* d_G6blk, hipStream_t s);
int launch_stiffness_generic(int P, int ncells, const int32_t* d_dofmap, const double* d_G6blk,
                             const double* d_D, const DMat& dm, double coeff, const double* d_x,
                             double* d_y, hipStream_t s);
int launch_stiffness_generic_u(int P, int ncells, const int32_t* d_uoff, const int32_t* d_uniq, const uint16_t* d_loc,
                               const double* d_G6blk, const double* d_D, const DMat& dm, double coeff,
                               const double* d_x, double* d_y, hipStream_t s);
int launch_stiffness_box(int P, int nx, int ny, int nz, int bx, int by, int bz, const double* d_G6blk,
                         const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                         hipStream_t s);
bool march_variant(int P, int variant, int* bx, int* by);
int launch_stiffness_march(int P, int variant, int nx, int ny, int nz, int lz, int lz0, const double* d_G6blk,
                           const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                           const int32_t* d_items, int nitems, hipStream_t s);
// indexed marching kernels for arbitrary dofmaps (generic_plan.cpp, stiffness_march_idx.hip, stiffness_march_ks.hip)
struct MarchPlan {
  bool ok = false;                      // false: the mesh does not tile into lattice columns
  int BX = 0, BY = 0, lz = 0, nitems = 0, npatterns = 0, tile_size = 0;
  int reoriented = 0, ncomponents = 0;  // cells looked at in a rotated / reflected frame; lattice components
  double fill = 0.0;                    // cells / cell slots
  std::vector<int32_t> slot_cell;       // [nitems * lz * BX * BY] cell index or -1, slot order [layer][ly][lx]
  std::vector<uint8_t> cell_orient;     // [ncells] orientation code (orient_decode)
  std::vector<int32_t> item_base;       // [nitems] smallest dof of the item
  std::vector<int32_t> item_pattern;    // [nitems]
  std::vector<int32_t> item_layers;     // [nitems] non-empty layers (<= lz)
  std::vector<std::array<int32_t, 4>> item_key;   // [nitems] (lattice component, column x, column y, z segment)
  std::vector<int32_t> pat_off;         // [npatterns][tile_size] dof - base per tile position, -1 = uncovered
};
struct MarchPlanDev {
  int nitems = 0, lz = 0, tile_size = 0, bx = 0, by = 0;
  int32_t *d_item_base = nullptr, *d_item_pattern = nullptr, *d_item_layers = nullptr, *d_pat_off = nullptr;
};
// cell orientations: lattice axis m of a cell runs along its own axis raw_axis[m], reversed when flip[m]
void orient_decode(int code, int raw_axis[3], int flip[3]);
int orient_local_index(int code, int n, int i, int j, int k);   // raw tensor index of lattice-frame node (i, j, k)
int orient_sign(int code);                                       // +1 rotation, -1 reflection
int build_march_plan(int P, size_t ncells, const int32_t* tdm, int BX, int BY, int lz_max, int lz_fixed, bool normalise,
                     MarchPlan* plan);
constexpr int OP_KIND_STIFFNESS = 0, OP_KIND_MASS = 1;
// column cross-section, LDS need and LDS budget (per workgroup) of the indexed marching kernel of (kind, P)
void march_idx_shape(int kind, int P, int* bx, int* by);   // stiffness: keeps a compiled (*bx, *by) of the k-split kernel
size_t march_idx_lds_bytes(int kind, int P, int BX, int BY, int lz);
size_t march_idx_lds_budget(int kind, int P, int BX, int BY);
// dense mass on the lattice columns (mass_march.hip)
void mass_march_shape(int P, int* bx, int* by);
size_t mass_march_lds_bytes(int P, int BX, int BY, int lz);
int launch_mass_march(int P, const MarchPlanDev& pd, const double* d_detJblk, const double* d_phi1, const double* d_x,
                      double* d_y, hipStream_t s);
int launch_stiffness_march_idx(int P, const MarchPlanDev& pd, const double* d_G6blk, const double* d_D,
                               const DMat& dm, double coeff, const double* d_x, double* d_y, const int32_t* d_items,
                               int nitems, hipStream_t s);
// k-split marching kernel (stiffness_march_ks.hip): the stiffness kernel of every degree
bool march_ks_shape(int P, int* bx, int* by);   // keeps a compiled (*bx, *by), else sets the degree's default
int march_ks_resident(int P, int bx, int by);   // workgroups resident on the chip
size_t march_ks_lds_bytes(int P, int BX, int BY, int lz, bool idx);
int launch_stiffness_march_ks_box(int P, int bx, int by, int nx, int ny, int nz, int lz, int lz0, const double* d_G6blk,
                                  const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                                  const int32_t* d_items, int nitems, hipStream_t s);
int launch_stiffness_march_ks_idx(int P, int bx, int by, const MarchPlanDev& pd, const double* d_G6blk, const double* d_D,
                                  const DMat& dm, double coeff, const double* d_x, double* d_y, const int32_t* d_items,
                                  int nitems, hipStream_t s);
// dense simplex operator (stiffness_dense.hip)
struct DenseOpData;
int dense_setup(int nd, int nq, int ncells, int ndofs, const int32_t* dofmap, const double* dphi,
                const double* weights, const double* xverts, const int32_t* geom_dofmap, DenseOpData** out);
void dense_free(DenseOpData* d);
size_t dense_bytes(const DenseOpData* d);
int launch_stiffness_dense(const DenseOpData* d, double coeff, int do_clamp, const double* d_x, double* d_y,
                           hipStream_t s);
int launch_mass_lumped(int64_t nentries, const int32_t* d_dofmap, const double* d_detJ, const double* d_x,
                       double* d_y, hipStream_t s);
int launch_mass_lumped_u(int ncells, int nd, int CB, const int32_t* d_uoff, const int32_t* d_uniq,
                         const uint16_t* d_loc, const double* d_detJ, const double* d_x, double* d_y, hipStream_t s);
int mass_dense_cells_per_batch(int mx);
int launch_mass_dense_col(int P, int ncells, const int32_t* d_uoff, const int32_t* d_uniq, const uint16_t* d_loc,
                          const double* d_phi1, const double* d_detJ, const double* d_x, double* d_y, hipStream_t s);
int launch_mass_dense(int P, int nq1, int ncells, const int32_t* d_dofmap, const int32_t* d_uoff,
                      const int32_t* d_uniq, const uint16_t* d_loc, int CBu, const double* d_phi1,
                      const double* d_detJ, const double* d_x, double* d_y, hipStream_t s);

}  // namespace wf
