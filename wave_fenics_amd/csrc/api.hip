// C ABI of libwavehip: device shims, geometry setup, operator handles.
// See include/wavehip.h for the reference interface each entry point replaces.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "common.h"

using namespace wf;

struct wf_op {
  int kind = 0, P = 0, n = 0, nd = 0, nq = 0, ncells = 0, ndofs = 0;
  int structured = 0, nx = 0, ny = 0, nz = 0, bx = 1, by = 1, bz = 1;
  int nq1 = 0;
  int march = 0, march_variant = 0, lz = 1;   // marching box kernel (stiffness_march.hip)
  int lz0_split = 1;                          // length of the first z segment of the interior / interface parts
  double coeff = 0.0;
  DMat dm{};
  int32_t* d_dofmap = nullptr;
  double* d_G6blk = nullptr;
  double* d_detJ = nullptr;
  double* d_D = nullptr;
  double* d_phi1 = nullptr;
  double* d_mdiag = nullptr;
  // batch-unique gather/scatter lists of the generic stiffness kernel
  int32_t* d_uoff = nullptr;
  int32_t* d_uniq = nullptr;
  uint16_t* d_loc = nullptr;
  int generic_unique = 0, unique_cb = 0, dense_square = 0;
  // work-item lists of the marching kernel: [0] interior, [1] interface, [2]/[3] the two halves of the interior
  int32_t* d_items[4] = {nullptr, nullptr, nullptr, nullptr};
  int nitems[4] = {0, 0, 0, 0};
  int have_parts = 0;
  MarchPlanDev plan{};            // lattice columns of the indexed marching kernel (stiffness_march_idx.hip)
  int have_plan = 0, plan_patterns = 0;
  DenseOpData* dense = nullptr;   // dense simplex operator (stiffness_dense.hip)
  int dense_clamp = 1;
  size_t device_bytes = 0;
  int kernel_id = WF_KERNEL_NONE;   // what wf_op_apply launches (wf_op_info_t.kernel)
  int plan_reoriented = 0;
  double plan_fill = 0.0;
  wf_tuning tun{};
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t count, size_t* total)
{
  *p = nullptr;
  if (count == 0) return WF_OK;
  WF_HIP_CHECK(hipMalloc((void**)p, count * sizeof(T)));
  if (total) *total += count * sizeof(T);
  return WF_OK;
}

template <typename T>
int dev_upload(T** p, const T* host, size_t count, size_t* total)
{
  int rc = dev_alloc(p, count, total);
  if (rc != WF_OK) return rc;
  if (count) WF_HIP_CHECK(hipMemcpy(*p, host, count * sizeof(T), hipMemcpyHostToDevice));
  return WF_OK;
}

// temporary device buffer freed at scope exit
template <typename T>
struct Scratch {
  T* p = nullptr;
  ~Scratch()
  {
    if (p) (void)hipFree(p);
  }
};

void free_op(wf_op* op)
{
  if (!op) return;
  (void)hipFree(op->d_dofmap);
  (void)hipFree(op->d_G6blk);
  (void)hipFree(op->d_detJ);
  (void)hipFree(op->d_D);
  (void)hipFree(op->d_phi1);
  (void)hipFree(op->d_mdiag);
  (void)hipFree(op->d_uoff);
  (void)hipFree(op->d_uniq);
  (void)hipFree(op->d_loc);
  for (int k = 0; k < 4; ++k) (void)hipFree(op->d_items[k]);
  (void)hipFree(op->plan.d_item_base);
  (void)hipFree(op->plan.d_item_pattern);
  (void)hipFree(op->plan.d_item_layers);
  (void)hipFree(op->plan.d_pat_off);
  dense_free(op->dense);
  delete op;
}

int upload_tables(int P, Scratch<double>& d_pts, Scratch<double>& d_wts)
{
  const int n = P + 1;
  std::vector<double> pts(n), wts(n);
  gll_points_weights(n, pts.data(), wts.data());
  int rc = dev_upload(&d_pts.p, pts.data(), n, nullptr);
  if (rc != WF_OK) return rc;
  return dev_upload(&d_wts.p, wts.data(), n, nullptr);
}

// Batch-unique gather/scatter lists: for every batch of CB consecutive cells the
// sorted list of its distinct dofs (uniq, offsets uoff) and the position of each
// element-local dof in that list (loc).  Kernels read x once per unique dof, sum
// the batch in LDS and issue one global atomic per unique dof.
int build_unique_lists(wf_op* op, size_t ncells, int nd, int CB)
{
  if (ncells == 0) return WF_OK;
  if ((size_t)CB * nd > 65535) {
    set_error("build_unique_lists: batch too large for 16-bit local indices");
    return WF_ERR_UNSUPPORTED;
  }
  const size_t nbatch = (ncells + CB - 1) / CB;
  std::vector<int32_t> tdm(ncells * nd);
  WF_HIP_CHECK(hipMemcpy(tdm.data(), op->d_dofmap, tdm.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int32_t> uoff(nbatch + 1, 0), uniq, tmp;
  std::vector<uint16_t> loc(ncells * nd);
  uniq.reserve(ncells * nd / 2);
  for (size_t b = 0; b < nbatch; ++b) {
    const size_t c0 = b * CB, nc = std::min<size_t>(CB, ncells - c0);
    tmp.assign(tdm.begin() + c0 * nd, tdm.begin() + (c0 + nc) * nd);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    for (size_t e = c0 * nd; e < (c0 + nc) * nd; ++e)
      loc[e] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), tdm[e]) - tmp.begin());
    uniq.insert(uniq.end(), tmp.begin(), tmp.end());
    uoff[b + 1] = (int32_t)uniq.size();
  }
  int rc;
  if ((rc = dev_upload(&op->d_uoff, uoff.data(), uoff.size(), &op->device_bytes)) != WF_OK) return rc;
  if ((rc = dev_upload(&op->d_uniq, uniq.data(), uniq.size(), &op->device_bytes)) != WF_OK) return rc;
  if ((rc = dev_upload(&op->d_loc, loc.data(), loc.size(), &op->device_bytes)) != WF_OK) return rc;
  op->generic_unique = 1;
  op->unique_cb = CB;
  return WF_OK;
}

constexpr int kKsVariant = 3;   // wf_tuning.variant - 1 == 3: the k-split marching kernel (stiffness_march_ks.hip)

void default_box_block(int P, const wf_tuning& tun, int* bx, int* by, int* bz)
{
  switch (P) {
    case 1: *bx = 4; *by = 4; *bz = 4; break;
    case 2: *bx = 3; *by = 3; *bz = 3; break;
    case 3: *bx = 4; *by = 2; *bz = 2; break;
    case 4: *bx = 5; *by = 2; *bz = 1; break;
    case 5: *bx = 7; *by = 1; *bz = 1; break;
    case 6: *bx = 5; *by = 1; *bz = 1; break;
    default: *bx = 4; *by = 1; *bz = 1; break;
  }
  if (tun.bx > 0 && tun.by > 0 && tun.bz > 0 && tun.bx * tun.by * tun.bz * (P + 1) * (P + 1) <= 256) {
    *bx = tun.bx;
    *by = tun.by;
    *bz = tun.bz;
  }
}

// kernel the batch path runs for each operator kind (wf_op_info_t.kernel)
int batch_kernel_id(const wf_op* op)
{
  switch (op->kind) {
    case WF_OP_STIFFNESS: return op->generic_unique ? WF_KERNEL_BATCH_UNIQUE : WF_KERNEL_ELEMENTWISE;
    case WF_OP_MASS_LUMPED:
      return op->d_mdiag ? WF_KERNEL_DIAGONAL : (op->generic_unique ? WF_KERNEL_BATCH_UNIQUE : WF_KERNEL_ELEMENTWISE);
    default: return (op->dense_square && op->generic_unique) ? WF_KERNEL_BATCH_UNIQUE : WF_KERNEL_MASS_DENSE_ANY;
  }
}

}  // namespace

extern "C" {

// ---- device runtime shims ------------------------------------------------
int wf_device_count(int* count)
{
  WF_REQUIRE(count != nullptr, "wf_device_count: null output");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    set_error(std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
    return WF_ERR_NODEVICE;
  }
  return WF_OK;
}

int wf_set_device(int device)
{
  int count = 0;
  int rc = wf_device_count(&count);
  if (rc != WF_OK) return rc;
  if (device < 0 || device >= count) {
    // utils.hpp:30-34: "The number of MPI processes should be less or equal the number of available devices"
    set_error("wf_set_device: device " + std::to_string(device) + " not available (" + std::to_string(count)
              + " devices)");
    return WF_ERR_NODEVICE;
  }
  WF_HIP_CHECK(hipSetDevice(device));
  return WF_OK;
}

int wf_device_info(int device, char* name, size_t name_len, size_t* total_mem, int* num_cu)
{
  hipDeviceProp_t prop;
  WF_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (name && name_len) {
    std::strncpy(name, prop.name, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (total_mem) *total_mem = prop.totalGlobalMem;
  if (num_cu) *num_cu = prop.multiProcessorCount;
  return WF_OK;
}

int wf_malloc(void** d_ptr, size_t bytes)
{
  WF_REQUIRE(d_ptr != nullptr, "wf_malloc: null output");
  *d_ptr = nullptr;
  if (bytes == 0) return WF_OK;
  WF_HIP_CHECK(hipMalloc(d_ptr, bytes));
  return WF_OK;
}
int wf_free(void* d_ptr)
{
  if (d_ptr) WF_HIP_CHECK(hipFree(d_ptr));
  return WF_OK;
}
int wf_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes)
{
  if (bytes) WF_HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return WF_OK;
}
int wf_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes)
{
  if (bytes) WF_HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return WF_OK;
}
int wf_memset(void* d_dst, int value, size_t bytes, void* stream)
{
  if (bytes) WF_HIP_CHECK(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
  return WF_OK;
}
int wf_sync(void* stream)
{
  if (stream)
    WF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  else
    WF_HIP_CHECK(hipDeviceSynchronize());
  return WF_OK;
}

// ---- geometry --------------------------------------------------------------
static int geometry_hex_rule(int n, const double* h_pts, const double* h_wts, int ncells, int nverts,
                             const double* h_xverts, const int32_t* h_geom_dofmap, int use_fabs, int clamp, double* h_G,
                             double* h_detJ, const char* who)
{
  if (!(ncells >= 0 && nverts >= 0 && h_xverts && h_geom_dofmap)) {
    set_error(std::string(who) + ": bad arguments");
    return WF_ERR_INVALID;
  }
  for (size_t e = 0; e < (size_t)ncells * 8; ++e)
    if (h_geom_dofmap[e] < 0 || h_geom_dofmap[e] >= nverts) {
      set_error(std::string(who) + ": vertex index out of range");
      return WF_ERR_INVALID;
    }
  const size_t nq = (size_t)n * n * n;
  Scratch<double> d_x, d_pts, d_wts, d_G, d_det;
  Scratch<int32_t> d_gd;
  int rc;
  if ((rc = dev_upload(&d_x.p, h_xverts, (size_t)nverts * 3, nullptr)) != WF_OK) return rc;
  if ((rc = dev_upload(&d_gd.p, h_geom_dofmap, (size_t)ncells * 8, nullptr)) != WF_OK) return rc;
  if ((rc = dev_upload(&d_pts.p, h_pts, (size_t)n, nullptr)) != WF_OK) return rc;
  if ((rc = dev_upload(&d_wts.p, h_wts, (size_t)n, nullptr)) != WF_OK) return rc;
  if (h_G && (rc = dev_alloc(&d_G.p, (size_t)ncells * nq * 9, nullptr)) != WF_OK) return rc;
  if (h_detJ && (rc = dev_alloc(&d_det.p, (size_t)ncells * nq, nullptr)) != WF_OK) return rc;
  // the kernel is generic in the number of points per direction (P + 1 = n)
  if ((rc = launch_geometry_hex(n - 1, ncells, d_x.p, d_gd.p, d_pts.p, d_wts.p, use_fabs, clamp, d_G.p, nullptr,
                                d_det.p, nullptr)) != WF_OK)
    return rc;
  WF_HIP_CHECK(hipDeviceSynchronize());
  if (h_G) WF_HIP_CHECK(hipMemcpy(h_G, d_G.p, (size_t)ncells * nq * 9 * sizeof(double), hipMemcpyDeviceToHost));
  if (h_detJ) WF_HIP_CHECK(hipMemcpy(h_detJ, d_det.p, (size_t)ncells * nq * sizeof(double), hipMemcpyDeviceToHost));
  return WF_OK;
}

int wf_geometry_hex(int P, int ncells, int nverts, const double* h_xverts, const int32_t* h_geom_dofmap,
                    int use_fabs, int clamp, double* h_G, double* h_detJ)
{
  if (P < 1 || P > kMaxDegree) {
    set_error("wf_geometry_hex: degree must be 1..7");
    return WF_ERR_UNSUPPORTED;
  }
  const int n = P + 1;
  std::vector<double> pts(n), wts(n);
  gll_points_weights(n, pts.data(), wts.data());
  return geometry_hex_rule(n, pts.data(), wts.data(), ncells, nverts, h_xverts, h_geom_dofmap, use_fabs, clamp, h_G,
                           h_detJ, "wf_geometry_hex");
}

int wf_geometry_hex_rule(int ncells, int nverts, const double* h_xverts, const int32_t* h_geom_dofmap, int nq1,
                         const double* h_points1, const double* h_weights1, int use_fabs, int clamp, double* h_G,
                         double* h_detJ)
{
  WF_REQUIRE(nq1 >= 1 && nq1 <= WF_MAX_QUAD_POINTS && h_points1 && h_weights1, "wf_geometry_hex_rule: bad rule");
  return geometry_hex_rule(nq1, h_points1, h_weights1, ncells, nverts, h_xverts, h_geom_dofmap, use_fabs, clamp, h_G,
                           h_detJ, "wf_geometry_hex_rule");
}

// ---- operators -------------------------------------------------------------
int wf_op_create(const wf_op_desc* desc, wf_op** out)
{
  WF_REQUIRE(desc && out, "wf_op_create: null argument");
  *out = nullptr;
  const int P = desc->degree;
  if (P < 1 || P > kMaxDegree) {
    set_error("wf_op_create: degree must be 1..7 (hexahedron)");   // mass.hpp:91-92 "Not implemented"
    return WF_ERR_UNSUPPORTED;
  }
  WF_REQUIRE(desc->kind == WF_OP_STIFFNESS || desc->kind == WF_OP_MASS_LUMPED || desc->kind == WF_OP_MASS_DENSE,
             "wf_op_create: unknown operator kind");
  WF_REQUIRE(desc->ncells >= 0 && desc->ndofs >= 0, "wf_op_create: negative size");
  WF_REQUIRE(desc->h_dofmap || desc->ncells == 0, "wf_op_create: dofmap missing");
  const int n = P + 1, nd = n * n * n;
  const size_t ncells = (size_t)desc->ncells;
  const bool have_mesh = desc->h_xverts && desc->h_geom_dofmap;

  // host-side validation of every index the kernels will dereference
  for (size_t e = 0; e < ncells * nd; ++e)
    WF_REQUIRE(desc->h_dofmap[e] >= 0 && desc->h_dofmap[e] < desc->ndofs, "wf_op_create: dofmap entry out of range");
  if (desc->h_perm) {
    std::vector<char> seen(nd, 0);
    for (int k = 0; k < nd; ++k) {
      WF_REQUIRE(desc->h_perm[k] >= 0 && desc->h_perm[k] < nd && !seen[desc->h_perm[k]],
                 "wf_op_create: perm is not a permutation");
      seen[desc->h_perm[k]] = 1;
    }
  }
  if (have_mesh)
    for (size_t e = 0; e < ncells * 8; ++e)
      WF_REQUIRE(desc->h_geom_dofmap[e] >= 0 && desc->h_geom_dofmap[e] < desc->nverts,
                 "wf_op_create: vertex index out of range");

  // Axis order of the caller's tensor indices.  The engine's is x-FASTEST: l = i + n (j + n k),
  // i along x.  With WF_FLAG_TENSOR_X_SLOWEST the caller's tensor index (the domain of h_perm
  // -- or of the dofmap itself when h_perm is NULL -- and the point index of h_G / h_detJ) is
  // l' = (i n + j) n + k, the order of Basix' tensor-product factorisation.  Both are folded
  // into one element permutation and one point permutation here; the 3x3 axes of G keep
  // their meaning (reference axes 0, 1, 2 = x, y, z in both conventions).
  const bool xslow = (desc->flags & WF_FLAG_TENSOR_X_SLOWEST) != 0;
  std::vector<int32_t> eff_perm;
  const int32_t* use_perm = desc->h_perm;
  if (xslow) {
    eff_perm.resize(nd);
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          const int lp = (i * n + j) * n + k;
          eff_perm[i + n * (j + n * k)] = desc->h_perm ? desc->h_perm[lp] : lp;
        }
    use_perm = eff_perm.data();
  }
  // point permutation of per-point input arrays: engine point q <- caller point qmap[q]
  auto make_qmap = [&](int m) {
    std::vector<int32_t> q((size_t)m * m * m);
    for (int k = 0; k < m; ++k)
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) q[i + m * (j + m * k)] = xslow ? (i * m + j) * m + k : i + m * (j + m * k);
    return q;
  };

  std::unique_ptr<wf_op, void (*)(wf_op*)> op(new wf_op, free_op);
  op->kind = desc->kind;
  op->P = P;
  op->n = n;
  op->nd = nd;
  op->nq = nd;
  op->ncells = desc->ncells;
  op->ndofs = desc->ndofs;
  op->coeff = -1.0 * desc->c0 * desc->c0;   // operators.hpp:115
  const wf_tuning tun = desc->tuning ? *desc->tuning : wf_tuning{};
  op->tun = tun;
  const bool force_batch = tun.kernel == WF_KERNEL_FORCE_BATCH || tun.kernel == WF_KERNEL_FORCE_ELEMENTWISE
                           || tun.kernel == WF_KERNEL_FORCE_MASS_ANY;
  const bool no_unique = tun.kernel == WF_KERNEL_FORCE_ELEMENTWISE;
  int rc;

  // ---- default path of the stiffness operator and of the dense mass with a square 1-D table: marching
  // over lattice columns found in the caller's mesh (generic_plan.cpp) --------------------------
  const bool plan_stiffness = desc->kind == WF_OP_STIFFNESS;
  const bool plan_mass = desc->kind == WF_OP_MASS_DENSE && desc->nq1 == n && desc->h_phi1
                         && (desc->h_detJ || (have_mesh && desc->h_qpts1 && desc->h_qwts1));
  // Dense mass with a COLLOCATED rule (the quadrature points are the nodes, phi1 = identity: the GLL rule of
  // demo/gpu_operator_monolithic/main.cpp:94-96 and of LinearGLL): Phi^T D Phi is the diagonal sum_cells det J w.
  // It is assembled once and applied as y += m .* x (24 B/dof) -- the result of the dense evaluation up to the
  // rounding of the six identity contractions.  A wf_tuning kernel hint keeps the dense kernels.
  if (plan_mass && tun.kernel == WF_KERNEL_AUTO && ncells > 0) {
    bool ident = true;
    for (int q = 0; q < n && ident; ++q)
      for (int a2 = 0; a2 < n; ++a2)
        if (std::abs(desc->h_phi1[q * n + a2] - (q == a2 ? 1.0 : 0.0)) > 1e-14) ident = false;
    if (ident) {
      std::vector<double> hd;
      const double* hsrc = desc->h_detJ;
      std::vector<int32_t> qm = make_qmap(n);
      if (!desc->h_detJ) {
        Scratch<double> d_x, d_qp, d_qw, d_det;
        Scratch<int32_t> d_gd;
        hd.resize(ncells * nd);
        const int use_fabs = (desc->flags & WF_FLAG_NO_FABS) ? 0 : 1;
        if ((rc = dev_upload(&d_x.p, desc->h_xverts, (size_t)desc->nverts * 3, nullptr)) != WF_OK) return rc;
        if ((rc = dev_upload(&d_gd.p, desc->h_geom_dofmap, ncells * 8, nullptr)) != WF_OK) return rc;
        if ((rc = dev_upload(&d_qp.p, desc->h_qpts1, (size_t)n, nullptr)) != WF_OK) return rc;
        if ((rc = dev_upload(&d_qw.p, desc->h_qwts1, (size_t)n, nullptr)) != WF_OK) return rc;
        if ((rc = dev_alloc(&d_det.p, ncells * nd, nullptr)) != WF_OK) return rc;
        if ((rc = launch_geometry_hex(P, desc->ncells, d_x.p, d_gd.p, d_qp.p, d_qw.p, use_fabs, 0, nullptr, nullptr, d_det.p,
                                      nullptr)) != WF_OK)
          return rc;
        WF_HIP_CHECK(hipDeviceSynchronize());
        WF_HIP_CHECK(hipMemcpy(hd.data(), d_det.p, hd.size() * sizeof(double), hipMemcpyDeviceToHost));
        hsrc = hd.data();
        for (int q = 0; q < nd; ++q) qm[q] = q;   // computed in the engine's point order
      }
      std::vector<double> md((size_t)desc->ndofs, 0.0);
      for (size_t c = 0; c < ncells; ++c)
        for (int l = 0; l < nd; ++l) {
          const int32_t dof = desc->h_dofmap[c * nd + (use_perm ? use_perm[l] : l)];   // tensor position l of cell c
          md[dof] += hsrc[c * nd + qm[l]];
        }
      if ((rc = dev_upload(&op->d_mdiag, md.data(), md.size(), &op->device_bytes)) != WF_OK) return rc;
      op->nq1 = n;
      op->nq = nd;
      op->kernel_id = WF_KERNEL_DIAGONAL;
      *out = op.release();
      return WF_OK;
    }
  }

  if ((plan_stiffness || plan_mass) && !force_batch && ncells > 0) {
    if (plan_stiffness)
      WF_REQUIRE(desc->h_G || have_mesh, "wf_op_create: stiffness needs h_G or the mesh (h_xverts, h_geom_dofmap)");
    const int pkind = plan_stiffness ? OP_KIND_STIFFNESS : OP_KIND_MASS;
    // tensor-ordered dofmap in the caller's cell order (permute.hpp:10-27)
    std::vector<int32_t> tdm;
    const int32_t* tsrc = desc->h_dofmap;
    if (use_perm) {
      tdm.resize(ncells * nd);
      if ((rc = wf_reorder_dofmap(desc->ncells, nd, use_perm, desc->h_dofmap, tdm.data())) != WF_OK) return rc;
      tsrc = tdm.data();
    }
    int BX = tun.bx, BY = tun.by;   // stiffness: a compiled cross-section of the k-split kernel, else the default
    march_idx_shape(pkind, P, &BX, &BY);
    const int CB = BX * BY, NTq = CB * n * n;
    // layers per work item: as many as the kernel's LDS budget per workgroup allows, at most 16
    int lz_max = plan_mass ? 32 : 16;   // (the dense-mass kernel streams its index table: no LDS limit)
    while (!plan_mass && lz_max > 1 && march_idx_lds_bytes(pkind, P, BX, BY, lz_max) > march_idx_lds_budget(pkind, P, BX, BY)) --lz_max;
    const int use_fabs = (desc->flags & WF_FLAG_NO_FABS) ? 0 : 1;
    const int clamp = (desc->flags & WF_FLAG_NO_CLAMP) ? 0 : 1;
    // A cell may be looked at with an axis reversed only if the 1-D table reads the same backwards,
    // phi1[n-1-q][n-1-a] == phi1[q][a] (true for every symmetric node / point set; the GLL derivative
    // matrix of the stiffness operator has the matching antisymmetry by construction).
    bool normalise = tun.orient == 0;
    if (plan_mass)
      for (int q = 0; q < n && normalise; ++q)
        for (int a2 = 0; a2 < n; ++a2)
          if (std::abs(desc->h_phi1[q * n + a2] - desc->h_phi1[(n - 1 - q) * n + (n - 1 - a2)]) > 1e-13) normalise = false;
    MarchPlan plan;
    if ((rc = build_march_plan(P, ncells, tsrc, BX, BY, lz_max, std::max(0, tun.lz), normalise, &plan)) != WF_OK) return rc;
    // mostly empty columns (a mesh one cell wide, a mesh shattered into tiny lattice components): the
    // marching kernel would read geometry for every slot -- batch kernel instead
    if (plan.ok && plan.fill < kMinPlanFill && tun.kernel != WF_KERNEL_FORCE_MARCH) plan.ok = false;
    if (plan.ok) {
      const int lz = plan.lz;
      const size_t nslots = (size_t)plan.nitems * lz * CB;
      op->plan.nitems = plan.nitems;
      op->plan.lz = lz;
      op->plan.tile_size = plan.tile_size;
      op->plan.bx = BX;
      op->plan.by = BY;
      op->bx = BX;
      op->by = BY;
      op->plan_patterns = plan.npatterns;
      op->plan_reoriented = plan.reoriented;
      op->plan_fill = plan.fill;
      if ((rc = dev_upload(&op->plan.d_item_base, plan.item_base.data(), plan.item_base.size(), &op->device_bytes)) != WF_OK) return rc;
      if ((rc = dev_upload(&op->plan.d_item_pattern, plan.item_pattern.data(), plan.item_pattern.size(), &op->device_bytes)) != WF_OK) return rc;
      if ((rc = dev_upload(&op->plan.d_item_layers, plan.item_layers.data(), plan.item_layers.size(), &op->device_bytes)) != WF_OK) return rc;
      if ((rc = dev_upload(&op->plan.d_pat_off, plan.pat_off.data(), plan.pat_off.size(), &op->device_bytes)) != WF_OK) return rc;
      op->kernel_id = WF_KERNEL_MARCH_IDX;
      // engine point index (lattice frame of the cell, x fastest) -> the caller's point index, per orientation
      std::vector<std::vector<int32_t>> pmaps(48);
      auto point_map = [&](int code) -> const std::vector<int32_t>& {
        auto& m = pmaps[code];
        if (m.empty()) {
          const std::vector<int32_t> qm = make_qmap(n);
          m.resize(nd);
          for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i) m[i + n * (j + n * k)] = qm[orient_local_index(code, n, i, j, k)];
        }
        return m;
      };

      if (plan_stiffness) {
        op->have_plan = 1;
        std::vector<double> D(2 * n * n);   // D, then its transpose (scalar-loaded by the k-split kernel)
        gll_derivative_matrix(P, D.data());
        for (int q = 0; q < n; ++q)
          for (int a2 = 0; a2 < n; ++a2) D[n * n + a2 * n + q] = D[q * n + a2];
        for (int q = 0; q < n * n; ++q) op->dm.v[q] = D[q];
        if ((rc = dev_upload(&op->d_D, D.data(), (size_t)2 * n * n, &op->device_bytes)) != WF_OK) return rc;

        // geometry in slot order [item][layer][ly][lx]; missing cells stay zero (they contribute nothing)
        const size_t g6 = nslots * nd * 6;
        if ((rc = dev_alloc(&op->d_G6blk, g6, &op->device_bytes)) != WF_OK) return rc;
        WF_HIP_CHECK(hipMemset(op->d_G6blk, 0, g6 * sizeof(double)));
        if (desc->h_G) {
          // G of a cell seen in the lattice frame: G'[a][b] = s_a s_b G[r_a][r_b] (r = the cell's own axis
          // along lattice axis a, s = -1 when reversed) at the relabelled point -- the operator
          // D^T G D is the same in every frame
          const size_t slab_slots = std::max<size_t>(CB, (((size_t)64 << 20) / (nd * 9 * sizeof(double))) / CB * CB);
          Scratch<double> d_G9;
          if ((rc = dev_alloc(&d_G9.p, std::min(slab_slots, nslots) * nd * 9, nullptr)) != WF_OK) return rc;
          std::vector<double> slab;
          for (size_t s0 = 0; s0 < nslots; s0 += slab_slots) {
            const size_t ns = std::min(slab_slots, nslots - s0);
            slab.assign(ns * nd * 9, 0.0);
            for (size_t q = 0; q < ns; ++q) {
              const int32_t c = plan.slot_cell[s0 + q];
              if (c < 0) continue;
              const int code = plan.cell_orient[c];
              const std::vector<int32_t>& pm = point_map(code);
              const double* gsrc = desc->h_G + (size_t)c * nd * 9;
              if (code == 0) {
                for (int pt = 0; pt < nd; ++pt) std::memcpy(&slab[(q * nd + pt) * 9], gsrc + (size_t)pm[pt] * 9, 9 * sizeof(double));
              } else {
                int ra[3], fl[3];
                orient_decode(code, ra, fl);
                for (int pt = 0; pt < nd; ++pt) {
                  const double* g9 = gsrc + (size_t)pm[pt] * 9;
                  double* dst = &slab[(q * nd + pt) * 9];
                  for (int a = 0; a < 3; ++a)
                    for (int b2 = 0; b2 < 3; ++b2) dst[a * 3 + b2] = ((fl[a] ^ fl[b2]) ? -1.0 : 1.0) * g9[ra[a] * 3 + ra[b2]];
                }
              }
            }
            WF_HIP_CHECK(hipMemcpy(d_G9.p, slab.data(), ns * nd * 9 * sizeof(double), hipMemcpyHostToDevice));
            if ((rc = launch_pack_G6(P, CB, (int)ns, d_G9.p, op->d_G6blk + (s0 / CB) * CB * nd * 6, nullptr)) != WF_OK) return rc;
            WF_HIP_CHECK(hipDeviceSynchronize());
          }
        } else {
          // one geometry thread per (present cell, point), written to the cell's slot; a cell is handed
          // over with its vertices relabelled into the lattice frame
          std::vector<int32_t> gd, slot_of;
          std::vector<uint8_t> sign;
          gd.reserve(ncells * 8);
          slot_of.reserve(ncells);
          sign.reserve(ncells);
          for (size_t q = 0; q < nslots; ++q) {
            const int32_t c = plan.slot_cell[q];
            if (c < 0) continue;
            const int code = plan.cell_orient[c];
            const int32_t* gsrc = desc->h_geom_dofmap + (size_t)c * 8;
            for (int v = 0; v < 8; ++v) gd.push_back(gsrc[orient_local_index(code, 2, v & 1, (v >> 1) & 1, (v >> 2) & 1)]);
            slot_of.push_back((int32_t)q);
            sign.push_back((uint8_t)(int8_t)orient_sign(code));
          }
          Scratch<double> d_x, d_pts, d_wts;
          Scratch<int32_t> d_gd, d_slot;
          Scratch<uint8_t> d_sign;
          if ((rc = dev_upload(&d_x.p, desc->h_xverts, (size_t)desc->nverts * 3, nullptr)) != WF_OK) return rc;
          if ((rc = dev_upload(&d_gd.p, gd.data(), gd.size(), nullptr)) != WF_OK) return rc;
          if ((rc = dev_upload(&d_slot.p, slot_of.data(), slot_of.size(), nullptr)) != WF_OK) return rc;
          if ((rc = dev_upload(&d_sign.p, sign.data(), sign.size(), nullptr)) != WF_OK) return rc;
          if ((rc = upload_tables(P, d_pts, d_wts)) != WF_OK) return rc;
          if ((rc = launch_geometry_hex_slots(P, CB, (int)slot_of.size(), d_x.p, d_gd.p, d_slot.p, d_sign.p, d_pts.p, d_wts.p,
                                              use_fabs, clamp, op->d_G6blk, nullptr)) != WF_OK)
            return rc;
        }
      } else {
        // det J * w per cell and point, host copy in the caller's cell order and point order
        std::vector<double> hd;
        const double* hsrc = desc->h_detJ;
        bool raw_points = false;   // hd is in the engine's (raw cell frame) point order already
        if (!desc->h_detJ) {
          Scratch<double> d_x, d_qp, d_qw, d_det;
          Scratch<int32_t> d_gd;
          hd.resize(ncells * nd);
          if ((rc = dev_upload(&d_x.p, desc->h_xverts, (size_t)desc->nverts * 3, nullptr)) != WF_OK) return rc;
          if ((rc = dev_upload(&d_gd.p, desc->h_geom_dofmap, ncells * 8, nullptr)) != WF_OK) return rc;
          if ((rc = dev_upload(&d_qp.p, desc->h_qpts1, (size_t)n, nullptr)) != WF_OK) return rc;
          if ((rc = dev_upload(&d_qw.p, desc->h_qwts1, (size_t)n, nullptr)) != WF_OK) return rc;
          if ((rc = dev_alloc(&d_det.p, ncells * nd, nullptr)) != WF_OK) return rc;
          if ((rc = launch_geometry_hex(P, desc->ncells, d_x.p, d_gd.p, d_qp.p, d_qw.p, use_fabs, 0, nullptr, nullptr, d_det.p,
                                        nullptr)) != WF_OK)
            return rc;
          WF_HIP_CHECK(hipDeviceSynchronize());
          WF_HIP_CHECK(hipMemcpy(hd.data(), d_det.p, hd.size() * sizeof(double), hipMemcpyDeviceToHost));
          hsrc = hd.data();
          raw_points = true;
        }
        // blocked slot layout [item * lz + layer][k][t], t = slot_in_layer * n^2 + j n + i; empty slots zero
        std::vector<double> blk(nslots * nd, 0.0);
        for (size_t q = 0; q < nslots; ++q) {
          const int32_t c = plan.slot_cell[q];
          if (c < 0) continue;
          const int code = plan.cell_orient[c];
          const size_t sub = q / CB, sl = q % CB;
          const double* src = hsrc + (size_t)c * nd;
          for (int k = 0; k < n; ++k)
            for (int ji = 0; ji < n * n; ++ji) {
              const int l = ji + n * n * k;
              const int rp = raw_points ? orient_local_index(code, n, l % n, (l / n) % n, l / (n * n)) : point_map(code)[l];
              blk[(sub * n + k) * NTq + sl * n * n + ji] = src[rp];
            }
        }
        // A non-symmetric 1-D table kept the caller's frames (normalise above).
        if ((rc = dev_upload(&op->d_detJ, blk.data(), blk.size(), &op->device_bytes)) != WF_OK) return rc;
        if ((rc = dev_upload(&op->d_phi1, desc->h_phi1, (size_t)n * n, &op->device_bytes)) != WF_OK) return rc;
        for (int q = 0; q < n * n; ++q) op->dm.v[q] = desc->h_phi1[q];
        op->have_plan = 2;
        op->nq1 = n;
        op->nq = nd;
        op->dense_square = 1;
      }
      WF_HIP_CHECK(hipDeviceSynchronize());
      *out = op.release();
      return WF_OK;
    }
    // the mesh does not tile into lattice columns: batch kernels below
  }

  // Internal cell order: cells are summed independently, so the operator may visit
  // them in any order.  Sorting by the smallest dof of each cell puts cells that
  // share dofs into the same workgroup batch whatever order the caller's mesh has
  // (a randomly ordered cfg2 mesh: 0.46 ms unsorted -> the 0.31 ms of the
  // lexicographic order).  wf_tuning.keep_cell_order keeps the caller's order.
  std::vector<int32_t> cperm(ncells);
  for (size_t c = 0; c < ncells; ++c) cperm[c] = (int32_t)c;
  if (!tun.keep_cell_order && ncells > 1) {
    std::vector<int32_t> key(ncells);
    for (size_t c = 0; c < ncells; ++c) key[c] = *std::min_element(desc->h_dofmap + c * nd, desc->h_dofmap + (c + 1) * nd);
    std::stable_sort(cperm.begin(), cperm.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  }
  bool identity_cells = true;
  for (size_t c = 0; c < ncells && identity_cells; ++c) identity_cells = cperm[c] == (int32_t)c;
  // permuted copies of the per-cell setup arrays (only when the order changes)
  std::vector<int32_t> p_geom;
  std::vector<double> p_detJ;
  const int32_t* h_geom_dofmap = desc->h_geom_dofmap;
  const double* h_detJ = desc->h_detJ;
  if (!identity_cells && have_mesh) {
    p_geom.resize(ncells * 8);
    for (size_t c = 0; c < ncells; ++c) std::memcpy(&p_geom[c * 8], desc->h_geom_dofmap + (size_t)cperm[c] * 8, 8 * sizeof(int32_t));
    h_geom_dofmap = p_geom.data();
  }
  if (desc->h_detJ && (!identity_cells || xslow)) {
    const int mq = desc->kind == WF_OP_MASS_DENSE ? desc->nq1 : n;
    WF_REQUIRE(mq >= 1 && mq <= 16, "wf_op_create: bad nq1");
    const size_t nqm = (size_t)mq * mq * mq;
    const std::vector<int32_t> qm = make_qmap(mq);
    p_detJ.resize(ncells * nqm);
    for (size_t c = 0; c < ncells; ++c) {
      const double* src = desc->h_detJ + (size_t)cperm[c] * nqm;
      for (size_t q = 0; q < nqm; ++q) p_detJ[c * nqm + q] = src[qm[q]];
    }
    h_detJ = p_detJ.data();
  }

  // tensor-ordered dofmap (permute.hpp:10-27 when the caller's element ordering differs)
  {
    std::vector<int32_t> tmp, tmp2;
    const int32_t* src = desc->h_dofmap;
    if (use_perm && ncells) {
      tmp.resize(ncells * nd);
      if ((rc = wf_reorder_dofmap(desc->ncells, nd, use_perm, desc->h_dofmap, tmp.data())) != WF_OK) return rc;
      src = tmp.data();
    }
    if (!identity_cells) {
      tmp2.resize(ncells * nd);
      for (size_t c = 0; c < ncells; ++c) std::memcpy(&tmp2[c * nd], src + (size_t)cperm[c] * nd, nd * sizeof(int32_t));
      src = tmp2.data();
    }
    if ((rc = dev_upload(&op->d_dofmap, src, ncells * nd, &op->device_bytes)) != WF_OK) return rc;
  }

  std::vector<double> D(2 * n * n);   // D, then its transpose (scalar-loaded by the k-split kernel)
  gll_derivative_matrix(P, D.data());
  for (int q = 0; q < n; ++q)
    for (int a2 = 0; a2 < n; ++a2) D[n * n + a2 * n + q] = D[q * n + a2];
  for (int q = 0; q < n * n; ++q) op->dm.v[q] = D[q];
  if ((rc = dev_upload(&op->d_D, D.data(), (size_t)2 * n * n, &op->device_bytes)) != WF_OK) return rc;

  const int use_fabs = (desc->flags & WF_FLAG_NO_FABS) ? 0 : 1;
  const int clamp = (desc->flags & WF_FLAG_NO_CLAMP) ? 0 : 1;
  Scratch<double> d_x, d_pts, d_wts;
  Scratch<int32_t> d_gd;
  if (have_mesh && ncells) {
    if ((rc = dev_upload(&d_x.p, desc->h_xverts, (size_t)desc->nverts * 3, nullptr)) != WF_OK) return rc;
    if ((rc = dev_upload(&d_gd.p, h_geom_dofmap, ncells * 8, nullptr)) != WF_OK) return rc;
    if ((rc = upload_tables(P, d_pts, d_wts)) != WF_OK) return rc;
  }

  if (desc->kind == WF_OP_STIFFNESS) {
    const int CB = cells_per_batch(P);
    const size_t nbatch = (ncells + CB - 1) / CB;
    // batch-unique dof lists (WF_KERNEL_FORCE_ELEMENTWISE keeps the element-wise scatter for comparison)
    if (!no_unique && (rc = build_unique_lists(op.get(), ncells, nd, CB)) != WF_OK) return rc;
    const size_t g6 = nbatch * CB * nd * 6;
    if ((rc = dev_alloc(&op->d_G6blk, g6, &op->device_bytes)) != WF_OK) return rc;
    if (g6) WF_HIP_CHECK(hipMemset(op->d_G6blk, 0, g6 * sizeof(double)));
    if (desc->h_G) {
      // reference layout G[ncells][nq][3][3] (precomputation.hpp:46) -> blocked upper triangle,
      // staged in slabs to bound the temporary
      const size_t slab_cells = std::max<size_t>(CB, (((size_t)64 << 20) / (nd * 9 * sizeof(double))) / CB * CB);
      Scratch<double> d_G9;
      if ((rc = dev_alloc(&d_G9.p, std::min(slab_cells, std::max<size_t>(ncells, 1)) * nd * 9, nullptr)) != WF_OK) return rc;
      std::vector<double> slab;
      for (size_t c0 = 0; c0 < ncells; c0 += slab_cells) {
        const size_t nc = std::min(slab_cells, ncells - c0);
        const double* hsrc = desc->h_G + c0 * nd * 9;
        if (!identity_cells || xslow) {
          const std::vector<int32_t> qm = make_qmap(n);
          slab.resize(nc * nd * 9);
          for (size_t c = 0; c < nc; ++c) {
            const double* gsrc = desc->h_G + (size_t)cperm[c0 + c] * nd * 9;
            for (int q = 0; q < nd; ++q) std::memcpy(&slab[(c * nd + q) * 9], gsrc + (size_t)qm[q] * 9, 9 * sizeof(double));
          }
          hsrc = slab.data();
        }
        WF_HIP_CHECK(hipMemcpy(d_G9.p, hsrc, nc * nd * 9 * sizeof(double), hipMemcpyHostToDevice));
        // slabs start on a batch boundary, so the packed destination is offset by whole batches
        if ((rc = launch_pack_G6(P, CB, (int)nc, d_G9.p, op->d_G6blk + (c0 / CB) * CB * nd * 6, nullptr)) != WF_OK) return rc;
        WF_HIP_CHECK(hipDeviceSynchronize());
      }
    } else if (have_mesh) {
      if ((rc = launch_geometry_hex(P, desc->ncells, d_x.p, d_gd.p, d_pts.p, d_wts.p, use_fabs, clamp, nullptr,
                                    op->d_G6blk, nullptr, nullptr)) != WF_OK)
        return rc;
    } else if (ncells) {
      set_error("wf_op_create: stiffness needs h_G or the mesh (h_xverts, h_geom_dofmap)");
      return WF_ERR_INVALID;
    }
  } else {
    // mass operators: detJ[ncells][nq]
    int nq1 = n;
    if (desc->kind == WF_OP_MASS_DENSE) {
      WF_REQUIRE(desc->h_phi1 && desc->nq1 >= 1 && desc->nq1 <= 16, "wf_op_create: dense mass needs phi1[nq1][P+1]");
      WF_REQUIRE(desc->h_detJ || (have_mesh && desc->h_qpts1 && desc->h_qwts1),
                 "wf_op_create: dense mass needs h_detJ[ncells][nq1^3] or the mesh and the 1-D rule (h_qpts1, h_qwts1)");
      nq1 = desc->nq1;
      if ((rc = dev_upload(&op->d_phi1, desc->h_phi1, (size_t)nq1 * n, &op->device_bytes)) != WF_OK) return rc;
    }
    op->nq1 = nq1;
    op->nq = nq1 * nq1 * nq1;
    {
      const int mx = std::max(n, nq1);
      // square tables (nq1 == P+1): column-thread kernel, batches of cells_per_batch(P)
      const bool square = desc->kind == WF_OP_MASS_DENSE && nq1 == n && tun.kernel != WF_KERNEL_FORCE_MASS_ANY;
      op->dense_square = square ? 1 : 0;
      const int CBm = (desc->kind == WF_OP_MASS_DENSE && !square) ? mass_dense_cells_per_batch(mx) : cells_per_batch(P);
      // dense mass: the unique-dof tile pays off only for small elements (measured at 10 M dofs:
      // P2 0.80 -> 0.68 ms, P4 0.43 -> 0.46 ms, P6 0.34 -> 0.41 ms)
      // lumped mass: the diagonal is pre-assembled below unless the caller asks for the
      // reference's element-wise sequence
      const bool elementwise = desc->kind == WF_OP_MASS_LUMPED && (desc->flags & WF_FLAG_MASS_ELEMENTWISE);
      const bool want = elementwise || (desc->kind == WF_OP_MASS_DENSE && (P <= 3 || square));
      if (want && !no_unique && (rc = build_unique_lists(op.get(), ncells, nd, CBm)) != WF_OK) return rc;
    }
    if (desc->h_detJ) {
      if ((rc = dev_upload(&op->d_detJ, h_detJ, ncells * op->nq, &op->device_bytes)) != WF_OK) return rc;
    } else if (have_mesh && desc->kind == WF_OP_MASS_DENSE) {
      // det J * w at the caller's tensor rule (precompute.hpp:49-116, mass.hpp:35-39)
      Scratch<double> d_qp, d_qw;
      if ((rc = dev_upload(&d_qp.p, desc->h_qpts1, (size_t)nq1, nullptr)) != WF_OK) return rc;
      if ((rc = dev_upload(&d_qw.p, desc->h_qwts1, (size_t)nq1, nullptr)) != WF_OK) return rc;
      if ((rc = dev_alloc(&op->d_detJ, ncells * op->nq, &op->device_bytes)) != WF_OK) return rc;
      if ((rc = launch_geometry_hex(nq1 - 1, desc->ncells, d_x.p, d_gd.p, d_qp.p, d_qw.p, use_fabs, 0, nullptr, nullptr,
                                    op->d_detJ, nullptr)) != WF_OK)
        return rc;
      WF_HIP_CHECK(hipDeviceSynchronize());
    } else if (have_mesh) {
      if ((rc = dev_alloc(&op->d_detJ, ncells * nd, &op->device_bytes)) != WF_OK) return rc;
      if ((rc = launch_geometry_hex(P, desc->ncells, d_x.p, d_gd.p, d_pts.p, d_wts.p, use_fabs, 0, nullptr, nullptr,
                                    op->d_detJ, nullptr)) != WF_OK)
        return rc;
    } else if (ncells) {
      set_error("wf_op_create: mass needs h_detJ or the mesh (h_xverts, h_geom_dofmap)");
      return WF_ERR_INVALID;
    }
    if (desc->kind == WF_OP_MASS_LUMPED && !(desc->flags & WF_FLAG_MASS_ELEMENTWISE)) {
      // A lumped mass is a diagonal: assemble m = M 1 once with the reference's own
      // sequence (gather 1, * detJ, scatter-add; spectral_mass.hpp:84-89) and apply it as
      // y += m .* x -- 24 B/dof instead of 8 nq + 4 nd per cell + 16 per dof.
      Scratch<double> d_ones;
      if ((rc = dev_alloc(&d_ones.p, (size_t)op->ndofs, nullptr)) != WF_OK) return rc;
      if ((rc = dev_alloc(&op->d_mdiag, (size_t)op->ndofs, &op->device_bytes)) != WF_OK) return rc;
      if (op->ndofs) {
        if ((rc = wf_fill(op->ndofs, 1.0, d_ones.p, nullptr)) != WF_OK) return rc;
        WF_HIP_CHECK(hipMemset(op->d_mdiag, 0, (size_t)op->ndofs * sizeof(double)));
        if (ncells && (rc = launch_mass_lumped((int64_t)ncells * nd, op->d_dofmap, op->d_detJ, d_ones.p, op->d_mdiag, nullptr)) != WF_OK)
          return rc;
        WF_HIP_CHECK(hipDeviceSynchronize());
      }
      op->device_bytes -= ncells * nd * sizeof(double);
      (void)hipFree(op->d_detJ);
      op->d_detJ = nullptr;
    }
  }
  WF_HIP_CHECK(hipDeviceSynchronize());
  op->kernel_id = batch_kernel_id(op.get());
  *out = op.release();
  return WF_OK;
}

int wf_op_create_box(int kind, int degree, int nx, int ny, int nz, const double* h_xverts, double c0, int flags,
                     wf_op** out)
{
  return wf_op_create_box_tuned(kind, degree, nx, ny, nz, h_xverts, c0, flags, nullptr, out);
}

int wf_op_create_box_tuned(int kind, int degree, int nx, int ny, int nz, const double* h_xverts, double c0, int flags,
                           const wf_tuning* tuning, wf_op** out)
{
  WF_REQUIRE(out != nullptr, "wf_op_create_box: null output");
  *out = nullptr;
  const int P = degree;
  if (P < 1 || P > kMaxDegree) {
    set_error("wf_op_create_box: degree must be 1..7 (hexahedron)");
    return WF_ERR_UNSUPPORTED;
  }
  WF_REQUIRE(kind == WF_OP_STIFFNESS || kind == WF_OP_MASS_LUMPED, "wf_op_create_box: kind must be stiffness or lumped mass");
  WF_REQUIRE(nx > 0 && ny > 0 && nz > 0 && h_xverts, "wf_op_create_box: bad mesh");
  const size_t NX = (size_t)P * nx + 1, NY = (size_t)P * ny + 1, NZ = (size_t)P * nz + 1;
  WF_REQUIRE(NX * NY * NZ < ((size_t)1 << 31), "wf_op_create_box: dof lattice exceeds int32");
  const int n = P + 1, nd = n * n * n;

  const wf_tuning tun = tuning ? *tuning : wf_tuning{};

  std::unique_ptr<wf_op, void (*)(wf_op*)> op(new wf_op, free_op);
  op->kind = kind;
  op->P = P;
  op->n = n;
  op->nd = nd;
  op->nq = nd;
  op->nq1 = n;
  op->ncells = nx * ny * nz;
  op->ndofs = (int)(NX * NY * NZ);
  op->structured = 1;
  op->nx = nx;
  op->ny = ny;
  op->nz = nz;
  op->coeff = -1.0 * c0 * c0;
  op->tun = tun;
  default_box_block(P, tun, &op->bx, &op->by, &op->bz);
  op->kernel_id = WF_KERNEL_DIAGONAL;
  if (kind == WF_OP_STIFFNESS) {
    // production kernel: marching columns (stiffness_march.hip; P >= 5: the k-split form,
    // stiffness_march_ks.hip).  wf_tuning: kernel = WF_KERNEL_FORCE_BOX_BLOCK selects the single-pass
    // block kernel, variant the compiled column cross-section, lz the layers per z segment.
    op->march = tun.kernel != WF_KERNEL_FORCE_BOX_BLOCK;
    op->kernel_id = op->march ? WF_KERNEL_MARCH_BOX : WF_KERNEL_BOX_BLOCK;
    if (op->march) {
      // P <= 4: the one-thread-per-column kernel (stiffness_march.hip), cross-section wf_tuning.variant - 1;
      // P >= 5: the k-split kernel (stiffness_march_ks.hip), cross-section wf_tuning.bx x by when compiled
      // (wf_tuning.variant = 4 selects it at P4 as well, for comparisons)
      static const int kDefaultVariant[8] = {0, 0, 0, 0, 1, kKsVariant, kKsVariant, kKsVariant};   // P4: 5x2 columns
      op->march_variant = tun.variant > 0 ? tun.variant - 1 : kDefaultVariant[P];
      if (P >= 5) op->march_variant = kKsVariant;
      if (op->march_variant == kKsVariant) {
        op->bx = tun.bx;
        op->by = tun.by;
        if (!march_ks_shape(P, &op->bx, &op->by)) {
          set_error("wf_op_create_box: the k-split kernel is compiled for degrees 4..7");
          return WF_ERR_UNSUPPORTED;
        }
      } else if (!march_variant(P, op->march_variant, &op->bx, &op->by)) {
        set_error("wf_op_create_box: wf_tuning.variant out of range");
        return WF_ERR_INVALID;
      }
      op->bz = 1;
      const int ncols = ((nx + op->bx - 1) / op->bx) * ((ny + op->by - 1) / op->by);
      // z segmentation: work items = columns x segments run in rounds of the 512
      // resident workgroups (2 per CU); each item pays ~1.5 layers of pipeline fill.
      // Pick the segment length that minimises rounds * (lz + 1.5).
      double best = 1e300;
      op->lz = nz;
      for (int nseg = 1; nseg <= nz; ++nseg) {
        const int lz = (nz + nseg - 1) / nseg;
        if (lz < 3 && nseg > 1) break;
        const long items = (long)ncols * ((nz + lz - 1) / lz);
        // workgroups per round
        const long resident = op->march_variant != kKsVariant ? 512 : march_ks_resident(P, op->bx, op->by);
        const double cost = (double)((items + resident - 1) / resident) * (lz + 1.5);
        if (cost < best - 1e-9) {
          best = cost;
          op->lz = lz;
        }
      }
      if (tun.lz > 0) op->lz = tun.lz;
    }
  }
  int rc;

  std::vector<double> D(2 * n * n);   // D, then its transpose (scalar-loaded by the k-split kernel)
  gll_derivative_matrix(P, D.data());
  for (int q = 0; q < n; ++q)
    for (int a2 = 0; a2 < n; ++a2) D[n * n + a2 * n + q] = D[q * n + a2];
  for (int q = 0; q < n * n; ++q) op->dm.v[q] = D[q];
  if ((rc = dev_upload(&op->d_D, D.data(), (size_t)2 * n * n, &op->device_bytes)) != WF_OK) return rc;

  Scratch<double> d_x, d_pts, d_wts;
  const size_t nverts = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
  if ((rc = dev_upload(&d_x.p, h_xverts, nverts * 3, nullptr)) != WF_OK) return rc;
  if ((rc = upload_tables(P, d_pts, d_wts)) != WF_OK) return rc;
  const int use_fabs = (flags & WF_FLAG_NO_FABS) ? 0 : 1;
  const int clamp = (flags & WF_FLAG_NO_CLAMP) ? 0 : 1;

  if (kind == WF_OP_STIFFNESS) {
    const size_t nblk = (size_t)((nx + op->bx - 1) / op->bx) * ((ny + op->by - 1) / op->by) * ((nz + op->bz - 1) / op->bz);
    const size_t g6 = nblk * op->bx * op->by * op->bz * nd * 6;
    if ((rc = dev_alloc(&op->d_G6blk, g6, &op->device_bytes)) != WF_OK) return rc;
    WF_HIP_CHECK(hipMemset(op->d_G6blk, 0, g6 * sizeof(double)));
    if ((rc = launch_geometry_box(P, nx, ny, nz, op->bx, op->by, op->bz, d_x.p, d_pts.p, d_wts.p, use_fabs, clamp,
                                  op->d_G6blk, nullptr, nullptr)) != WF_OK)
      return rc;
  } else {
    // pre-assembled lumped mass diagonal: y += m .* x is 24 B/dof instead of the
    // 34.8 B/dof gather/transform/scatter of spectral_mass.hpp:84-89
    if ((rc = dev_alloc(&op->d_mdiag, (size_t)op->ndofs, &op->device_bytes)) != WF_OK) return rc;
    WF_HIP_CHECK(hipMemset(op->d_mdiag, 0, (size_t)op->ndofs * sizeof(double)));
    if ((rc = launch_geometry_box(P, nx, ny, nz, 1, 1, 1, d_x.p, d_pts.p, d_wts.p, use_fabs, clamp, nullptr,
                                  op->d_mdiag, nullptr)) != WF_OK)
      return rc;
  }
  WF_HIP_CHECK(hipDeviceSynchronize());
  *out = op.release();
  return WF_OK;
}

int wf_op_create_dense_simplex(const wf_dense_desc* desc, wf_op** out)
{
  WF_REQUIRE(desc && out, "wf_op_create_dense_simplex: null argument");
  *out = nullptr;
  WF_REQUIRE(desc->nd > 0 && desc->nq > 0 && desc->ncells >= 0 && desc->ndofs >= 0, "wf_op_create_dense_simplex: bad sizes");
  WF_REQUIRE(desc->h_dofmap && desc->h_dphi && desc->h_weights && desc->h_xverts && desc->h_geom_dofmap,
             "wf_op_create_dense_simplex: null array");
  for (size_t e = 0; e < (size_t)desc->ncells * desc->nd; ++e)
    WF_REQUIRE(desc->h_dofmap[e] >= 0 && desc->h_dofmap[e] < desc->ndofs, "wf_op_create_dense_simplex: dofmap entry out of range");
  for (size_t e = 0; e < (size_t)desc->ncells * 4; ++e)
    WF_REQUIRE(desc->h_geom_dofmap[e] >= 0 && desc->h_geom_dofmap[e] < desc->nverts,
               "wf_op_create_dense_simplex: vertex index out of range");
  std::unique_ptr<wf_op, void (*)(wf_op*)> op(new wf_op, free_op);
  op->kind = WF_OP_STIFFNESS;
  op->P = 0;
  op->nd = desc->nd;
  op->nq = desc->nq;
  op->ncells = desc->ncells;
  op->ndofs = desc->ndofs;
  op->coeff = -1.0 * desc->c0 * desc->c0;
  op->dense_clamp = (desc->flags & WF_FLAG_NO_CLAMP) ? 0 : 1;
  int rc = dense_setup(desc->nd, desc->nq, desc->ncells, desc->ndofs, desc->h_dofmap, desc->h_dphi, desc->h_weights,
                       desc->h_xverts, desc->h_geom_dofmap, &op->dense);
  if (rc != WF_OK) return rc;
  op->device_bytes = dense_bytes(op->dense);
  // probe that the shape is compiled before handing the handle out
  *out = op.release();
  return WF_OK;
}

// the box marching kernels (one thread per column at P <= 4, two at P >= 5)
static int launch_box_march(const wf_op* op, int lz0, const double* d_x, double* d_y, const int32_t* d_items, int nitems,
                            hipStream_t s)
{
  if (op->march_variant == kKsVariant)
    return launch_stiffness_march_ks_box(op->P, op->bx, op->by, op->nx, op->ny, op->nz, op->lz, lz0, op->d_G6blk, op->d_D, op->dm, op->coeff,
                                         d_x, d_y, d_items, nitems, s);
  return launch_stiffness_march(op->P, op->march_variant, op->nx, op->ny, op->nz, op->lz, lz0, op->d_G6blk, op->d_D, op->dm,
                                op->coeff, d_x, d_y, d_items, nitems, s);
}

int wf_op_apply(wf_op* op, const double* d_x, double* d_y, void* stream)
{
  WF_REQUIRE(op && d_x && d_y, "wf_op_apply: null argument");
  MarkerScope mk("wf_op_apply");
  hipStream_t s = (hipStream_t)stream;
  if (op->dense) return launch_stiffness_dense(op->dense, op->coeff, op->dense_clamp, d_x, d_y, s);
  if (op->structured) {
    if (op->kind == WF_OP_STIFFNESS && op->march) return launch_box_march(op, op->lz, d_x, d_y, nullptr, 0, s);
    if (op->kind == WF_OP_STIFFNESS)
      return launch_stiffness_box(op->P, op->nx, op->ny, op->nz, op->bx, op->by, op->bz, op->d_G6blk, op->d_D, op->dm,
                                  op->coeff, d_x, d_y, s);
    return wf_pointwise_mult_add(op->ndofs, op->d_mdiag, d_x, d_y, stream);
  }
  switch (op->kind) {
    case WF_OP_STIFFNESS:
      if (op->have_plan == 1)
        return launch_stiffness_march_idx(op->P, op->plan, op->d_G6blk, op->d_D, op->dm, op->coeff, d_x, d_y, nullptr, 0, s);
      if (op->generic_unique)
        return launch_stiffness_generic_u(op->P, op->ncells, op->d_uoff, op->d_uniq, op->d_loc, op->d_G6blk, op->d_D,
                                          op->dm, op->coeff, d_x, d_y, s);
      return launch_stiffness_generic(op->P, op->ncells, op->d_dofmap, op->d_G6blk, op->d_D, op->dm, op->coeff, d_x,
                                      d_y, s);
    case WF_OP_MASS_LUMPED:
      if (op->d_mdiag) return wf_pointwise_mult_add(op->ndofs, op->d_mdiag, d_x, d_y, stream);
      if (op->generic_unique)
        return launch_mass_lumped_u(op->ncells, op->nd, op->unique_cb, op->d_uoff, op->d_uniq, op->d_loc, op->d_detJ, d_x,
                                    d_y, s);
      return launch_mass_lumped((int64_t)op->ncells * op->nd, op->d_dofmap, op->d_detJ, d_x, d_y, s);
    case WF_OP_MASS_DENSE:
      if (op->d_mdiag) return wf_pointwise_mult_add(op->ndofs, op->d_mdiag, d_x, d_y, stream);   // collocated rule
      if (op->have_plan == 2)
        return launch_mass_march(op->P, op->plan, op->d_detJ, op->d_phi1, d_x, d_y, s);
      if (op->dense_square && op->generic_unique)
        return launch_mass_dense_col(op->P, op->ncells, op->d_uoff, op->d_uniq, op->d_loc, op->d_phi1, op->d_detJ, d_x, d_y,
                                     s);
      return launch_mass_dense(op->P, op->nq1, op->ncells, op->d_dofmap, op->generic_unique ? op->d_uoff : nullptr,
                               op->d_uniq, op->d_loc, op->unique_cb, op->d_phi1, op->d_detJ, d_x, d_y, s);
  }
  set_error("wf_op_apply: corrupt handle");
  return WF_ERR_INVALID;
}

// uploads the interior / interface work-item lists (and the two interior halves)
static int set_item_lists(wf_op* op, std::vector<int32_t> (&items)[4])
{
  // the interior halves let a caller hide BOTH halo directions: forward update under
  // half A, reverse update under half B (alternate items so both halves span the mesh)
  items[2].clear();
  items[3].clear();
  for (size_t q = 0; q < items[0].size(); ++q) items[2 + (q & 1)].push_back(items[0][q]);
  for (int k = 0; k < 4; ++k) {
    if (op->d_items[k]) op->device_bytes -= (size_t)op->nitems[k] * sizeof(int32_t);
    (void)hipFree(op->d_items[k]);
    op->d_items[k] = nullptr;
    op->nitems[k] = (int)items[k].size();
    if (op->nitems[k]) {
      int rc = dev_upload(&op->d_items[k], items[k].data(), items[k].size(), &op->device_bytes);
      if (rc != WF_OK) return rc;
    }
  }
  op->have_parts = 1;
  return WF_OK;
}

int wf_op_set_ghost_faces(wf_op* op, int ghost_x0, int ghost_y0, int ghost_z0)
{
  WF_REQUIRE(op != nullptr, "wf_op_set_ghost_faces: null handle");
  if (!(op->structured && op->kind == WF_OP_STIFFNESS && op->march)) {
    set_error("wf_op_set_ghost_faces: only the marching box stiffness operator has lattice faces (wf_op_set_ghost_dofs "
              "splits any marching operator)");
    return WF_ERR_UNSUPPORTED;
  }
  const int nbx = (op->nx + op->bx - 1) / op->bx, nby = (op->ny + op->by - 1) / op->by;
  // With a ghost plane below, the first z segment is kept short (wf_tuning.lz0, default 3 layers):
  // only its first layer reads the ghost plane, but the whole segment has to wait for the halo, and
  // the less interface work there is the earlier the reverse exchange can start under the interior.
  op->lz0_split = op->lz;
  if (ghost_z0) op->lz0_split = std::max(1, std::min(op->tun.lz0 > 0 ? op->tun.lz0 : 3, op->lz));
  const int ncols = nbx * nby, nseg = 1 + (std::max(op->nz - op->lz0_split, 0) + op->lz - 1) / op->lz;
  std::vector<int32_t> items[4];
  for (int seg = 0; seg < nseg; ++seg)
    for (int col = 0; col < ncols; ++col) {
      const int Bx = col % nbx, By = col / nbx;
      // a work item is "interface" when it reads a ghost plane of x / adds into a ghost plane of y
      const bool iface = (ghost_x0 && Bx == 0) || (ghost_y0 && By == 0) || (ghost_z0 && seg == 0);
      items[iface ? 1 : 0].push_back(col + ncols * seg);
    }
  return set_item_lists(op, items);
}

int wf_op_set_ghost_dofs(wf_op* op, const int32_t* h_ghost_positions, int32_t nghosts)
{
  WF_REQUIRE(op != nullptr && nghosts >= 0 && (nghosts == 0 || h_ghost_positions), "wf_op_set_ghost_dofs: bad argument");
  const bool box = op->structured && op->kind == WF_OP_STIFFNESS && op->march;
  const bool idx = !op->structured && op->kind == WF_OP_STIFFNESS && op->have_plan == 1;
  if (!box && !idx) {
    set_error("wf_op_set_ghost_dofs: only the marching stiffness operators split into interior / interface work items "
              "(this operator runs a batch kernel)");
    return WF_ERR_UNSUPPORTED;
  }
  std::vector<char> ghost((size_t)op->ndofs, 0);
  for (int32_t g = 0; g < nghosts; ++g) {
    WF_REQUIRE(h_ghost_positions[g] >= 0 && h_ghost_positions[g] < op->ndofs, "wf_op_set_ghost_dofs: ghost position out of range");
    ghost[h_ghost_positions[g]] = 1;
  }
  const int P = op->P;
  std::vector<int32_t> items[4];
  if (box) {
    const int NX = P * op->nx + 1, NY = P * op->ny + 1;
    const size_t plane = (size_t)NX * NY;
    // a z ghost plane below shortens the first segment (see wf_op_set_ghost_faces)
    bool gz = false;
    for (size_t g = 0; g < plane && !gz; ++g) gz = ghost[g] != 0;
    op->lz0_split = op->lz;
    if (gz) op->lz0_split = std::max(1, std::min(op->tun.lz0 > 0 ? op->tun.lz0 : 3, op->lz));
    const int nbx = (op->nx + op->bx - 1) / op->bx, nby = (op->ny + op->by - 1) / op->by;
    const int ncols = nbx * nby, nseg = 1 + (std::max(op->nz - op->lz0_split, 0) + op->lz - 1) / op->lz;
    for (int seg = 0; seg < nseg; ++seg) {
      const int z0 = seg == 0 ? 0 : op->lz0_split + (seg - 1) * op->lz;
      const int z1 = std::min(op->nz, seg == 0 ? op->lz0_split : z0 + op->lz);
      for (int col = 0; col < ncols; ++col) {
        const int Bx = col % nbx, By = col / nbx;
        const int I0 = P * Bx * op->bx, J0 = P * By * op->by;
        const int I1 = std::min(NX - 1, I0 + P * op->bx), J1 = std::min(NY - 1, J0 + P * op->by);
        bool iface = false;
        for (int K = P * z0; K <= P * z1 && !iface; ++K)
          for (int J = J0; J <= J1 && !iface; ++J) {
            const char* row = &ghost[(size_t)I0 + (size_t)NX * J + plane * K];
            for (int I = 0; I <= I1 - I0; ++I)
              if (row[I]) {
                iface = true;
                break;
              }
          }
        items[iface ? 1 : 0].push_back(col + ncols * seg);
      }
    }
  } else {
    // an item is interface iff its dof tile (base + pattern offsets) contains a ghost position
    const int nit = op->plan.nitems;
    const size_t tsize = (size_t)op->plan.tile_size;
    std::vector<int32_t> base(nit), pat(nit), pat_off((size_t)op->plan_patterns * tsize);
    WF_HIP_CHECK(hipMemcpy(base.data(), op->plan.d_item_base, (size_t)nit * sizeof(int32_t), hipMemcpyDeviceToHost));
    WF_HIP_CHECK(hipMemcpy(pat.data(), op->plan.d_item_pattern, (size_t)nit * sizeof(int32_t), hipMemcpyDeviceToHost));
    WF_HIP_CHECK(hipMemcpy(pat_off.data(), op->plan.d_pat_off, pat_off.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int it = 0; it < nit; ++it) {
      const int32_t* off = &pat_off[(size_t)pat[it] * tsize];
      bool iface = false;
      for (size_t e = 0; e < tsize; ++e)
        if (off[e] >= 0 && ghost[(size_t)base[it] + off[e]]) {
          iface = true;
          break;
        }
      items[iface ? 1 : 0].push_back(it);
    }
  }
  return set_item_lists(op, items);
}

int wf_op_apply_part(wf_op* op, const double* d_x, double* d_y, int part, void* stream)
{
  WF_REQUIRE(op && d_x && d_y, "wf_op_apply_part: null argument");
  if (part == WF_PART_ALL) return wf_op_apply(op, d_x, d_y, stream);
  WF_REQUIRE(part >= WF_PART_INTERIOR && part <= WF_PART_INTERIOR_B, "wf_op_apply_part: unknown part");
  if (!op->have_parts) {
    set_error("wf_op_apply_part: call wf_op_set_ghost_dofs / wf_op_set_ghost_faces first");
    return WF_ERR_INVALID;
  }
  const int k = part - 1;   // WF_PART_INTERIOR, _INTERFACE, _INTERIOR_A, _INTERIOR_B
  if (op->nitems[k] == 0) return WF_OK;
  static const char* kPartName[4] = {"wf_op_apply_part interior", "wf_op_apply_part interface", "wf_op_apply_part interior A",
                                     "wf_op_apply_part interior B"};
  MarkerScope mk(kPartName[k]);
  if (op->structured) return launch_box_march(op, op->lz0_split, d_x, d_y, op->d_items[k], op->nitems[k], (hipStream_t)stream);
  return launch_stiffness_march_idx(op->P, op->plan, op->d_G6blk, op->d_D, op->dm, op->coeff, d_x, d_y, op->d_items[k],
                                    op->nitems[k], (hipStream_t)stream);
}

int wf_op_info(const wf_op* op, wf_op_info_t* info)
{
  WF_REQUIRE(op && info, "wf_op_info: null argument");
  info->kind = op->kind;
  info->degree = op->P;
  info->num_cells = op->ncells;
  info->num_dofs_cell = op->nd;
  info->num_quads = op->nq;
  info->ndofs = op->ndofs;
  info->structured = op->structured;
  info->flops = 4.0 * op->ncells * (double)op->nq * op->nd;   // mass.hpp:71
  if (op->dense) {
    info->flops = 12.0 * op->ncells * (double)op->nq * op->nd;                                  // dense skernel, SURVEY 8a3
    info->alg_bytes = (double)op->ncells * (48.0 + 4.0 * op->nd) + 16.0 * op->ndofs;
  } else if (op->kind == WF_OP_STIFFNESS)
    info->alg_bytes = (double)op->ncells * (48.0 * op->nq + 4.0 * op->nd) + 16.0 * op->ndofs;
  else if (op->d_mdiag)
    info->alg_bytes = 24.0 * op->ndofs;   // pre-assembled diagonal: read m, x, y + write y (SURVEY 8d counts 24)
  else
    info->alg_bytes = (double)op->ncells * (8.0 * op->nq + 4.0 * op->nd) + 16.0 * op->ndofs;
  info->device_bytes = op->device_bytes;
  info->items_interior = op->nitems[0];
  info->items_interface = op->nitems[1];
  info->kernel = op->dense ? WF_KERNEL_DENSE_SIMPLEX : op->kernel_id;
  info->plan_items = op->have_plan ? op->plan.nitems : 0;
  info->plan_patterns = op->have_plan ? op->plan_patterns : 0;
  info->plan_lz = op->have_plan ? op->plan.lz : (op->structured && op->march ? op->lz : 0);
  info->plan_reoriented = op->plan_reoriented;
  info->plan_fill = op->plan_fill;
  return WF_OK;
}

int wf_op_destroy(wf_op* op)
{
  free_op(op);
  return WF_OK;
}

}  // extern "C"
