/*
 * wavehip.h -- C ABI of libwavehip, the MI355X (gfx950) matrix-free operator
 * engine for the explicit-RK wave loop of Excalibur-SLE/wave-fenics.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types,
 * no exceptions.  Every entry point returns 0 on success or a negative
 * wf_status; wf_last_error() returns the message of the last failure on the
 * calling thread.  Each declaration cites the reference interface it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - fp64 data, int32 indices.
 *   - "h_" parameters are host pointers read during the call (setup);
 *     "d_" parameters are device pointers (hipMalloc / torch tensors).
 *   - apply is y += A x (the reference accumulates, operators.hpp:196-198,
 *     scatter.cu:43; the caller zeroes y, LinearGLL.hpp:173).
 *   - stream is a hipStream_t passed as void* (NULL = default stream); all
 *     device work is stream-ordered and asynchronous, nothing synchronises
 *     except wf_sync and the setup calls that read host memory.
 *   - element-local tensor ordering l = i + n*(j + n*k), n = P+1, i along x (x FASTEST).
 *     This is the order of every per-dof / per-point array handed over: the tensor side
 *     of h_perm, h_dofmap when h_perm is NULL, the point index q of h_G [c][q][3][3] and
 *     h_detJ [c][q], and the vertex index v = a + 2b + 4c of h_geom_dofmap.  A caller whose
 *     tensor order is x SLOWEST (Basix' get_tensor_product_representation, whose perm the
 *     reference uses as is, common/permute.hpp:12-17, spectral_mass.hpp:36-38) sets
 *     WF_FLAG_TENSOR_X_SLOWEST; pairing an x-slowest index with this engine's x-fastest
 *     geometry silently mixes G_xx with G_zz on any non-cubic cell.
 */
#ifndef WAVEHIP_H
#define WAVEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  WF_OK = 0,
  WF_ERR_INVALID = -1,     /* bad argument                                      */
  WF_ERR_UNSUPPORTED = -2, /* e.g. degree outside 1..7 (mass.hpp:91-92 throws)  */
  WF_ERR_HIP = -3,         /* a HIP runtime call failed (array.hpp:15-17 throws) */
  WF_ERR_NODEVICE = -4,    /* fewer devices than requested (utils.hpp:30-34)     */
  WF_ERR_COMM = -5         /* RCCL missing or a communication call failed
                              (the reference asserts on MPI status, VectorUpdater.hpp:120) */
} wf_status;

typedef struct wf_op wf_op;

const char* wf_last_error(void);
const char* wf_version(void);

/* ---- device runtime shims: common/cuda/utils.hpp:22-56, array.hpp:8-51 ---- */
int wf_device_count(int* count);
int wf_set_device(int device);                       /* utils::set_device      */
int wf_device_info(int device, char* name, size_t name_len, size_t* total_mem,
                   int* num_cu);                     /* utils::output_device_info */
int wf_malloc(void** d_ptr, size_t bytes);           /* cuda::array ctor       */
int wf_free(void* d_ptr);                            /* cuda::array dtor       */
int wf_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes); /* array::set */
int wf_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes); /* copy_to_host */
int wf_memset(void* d_dst, int value, size_t bytes, void* stream);
int wf_sync(void* stream);                           /* cudaDeviceSynchronize  */

/* ---- a1/a13: tabulation (host) ------------------------------------------
 * common/operators.hpp:13-32 tabulate_basis_and_permutation,
 * common/precompute.hpp:179-189 tabulate_1d.
 * 1-D GLL rule with P+1 points on [0,1] and the collocation derivative matrix
 * D[q*n + a] = l_a'(xi_q), clamped to -1/0/1 like the reference's dense table.
 * Any output pointer may be NULL. */
int wf_tabulate_gll(int P, double* h_points, double* h_weights, double* h_D);

/* 1-D quadrature rules as basix::quadrature::make_quadrature(type, interval, degree)
 * selects them (common/precompute.hpp:183-184, demo/gpu_operator/main.cpp:96-99,
 * common/operators.hpp:19): Gauss-Jacobi (alpha = beta = 0, i.e. Gauss-Legendre) with
 * (degree+2)/2 points, GLL with (degree+4)/2 points; on [0,1], ascending.
 * *npts is always set; h_points / h_weights (capacity WF_MAX_QUAD_POINTS) may be NULL. */
typedef enum { WF_QUAD_GLL = 0, WF_QUAD_GAUSS_JACOBI = 1 } wf_quadrature_type;
typedef enum { WF_VARIANT_GLL_WARPED = 0, WF_VARIANT_EQUISPACED = 1 } wf_lagrange_variant;
#define WF_MAX_QUAD_POINTS 16
int wf_quadrature_1d(int type, int degree, int* npts, double* h_points, double* h_weights);

/* common/precompute.hpp:179-189 tabulate_1d: the degree-P Lagrange basis of the given
 * variant on the interval at npts points; h_table[q*(P+1) + a] = l_a(x_q) (derivative 0)
 * or l_a'(x_q) (derivative 1).  Not clamped (neither is the reference's). */
int wf_tabulate_1d(int P, int variant, int npts, const double* h_points, int derivative, double* h_table);

/* Dense reference-layout table[4][nq][nd] (0 = values, 1..3 = d/dx,d/dy,d/dz),
 * nq = nd = (P+1)^3, tensor ordering, clamped (operators.hpp:23-29). */
int wf_tabulate_dense(int P, double* h_table);

/* common/permute.hpp:10-27 reorder_dofmap: out[c*nd + k] = in[c*nd + perm[k]]. */
int wf_reorder_dofmap(int ncells, int nd, const int32_t* h_perm,
                      const int32_t* h_in, int32_t* h_out);

/* Setup-time renumbering option: a dof numbering that follows the lattice columns the marching kernels walk
 * (work items in z-segment / column order, inside an item plane by plane, row by row), for spaces whose own
 * numbering scatters a cell's dofs over memory.  The reference takes DOLFINx's numbering as it comes
 * (fem::create_functionspace, demo/cpu_planar3d/main.cpp:47-52, reordered by DOLFINx's graph reordering); a
 * caller applies h_new_of_old to its dofmap and vectors before creating operators.  Host only.
 * h_dofmap: tensor-ordered (x fastest) [ncells][(degree+1)^3]; h_new_of_old[ndofs]. */
int wf_lattice_numbering(int degree, int64_t ncells, int32_t ndofs, const int32_t* h_dofmap, int32_t* h_new_of_old);

/* ---- a2/a13: geometry (device kernel, host in/out) -----------------------
 * common/precomputation.hpp:18-110 precompute_geometric_data (use_fabs = 1,
 * clamp = 1) and common/precompute.hpp:49-176 (use_fabs = 0, clamp = 0).
 * h_xverts [nverts][3], h_geom_dofmap [ncells][8] (vertex v = a + 2b + 4c).
 * Outputs in the reference layout: h_G [ncells][nq][3][3] (may be NULL),
 * h_detJ [ncells][nq] (may be NULL); quadrature = (P+1)^3 GLL points. */
int wf_geometry_hex(int P, int ncells, int nverts, const double* h_xverts,
                    const int32_t* h_geom_dofmap, int use_fabs, int clamp,
                    double* h_G, double* h_detJ);

/* The generic helpers of common/precompute.hpp:49-176 (compute_jacobian, _determinant,
 * _inverse, compute_geometrical_factor) at an arbitrary tensor-product rule
 * nq1 x nq1 x nq1 (point q = i + nq1*(j + nq1*k), weight w_i w_j w_k):
 * h_detJ [ncells][nq1^3] = det J * w (|det J| * w with use_fabs), h_G [ncells][nq1^3][3][3]
 * = K K^T detJ w.  Either output may be NULL. */
int wf_geometry_hex_rule(int ncells, int nverts, const double* h_xverts, const int32_t* h_geom_dofmap, int nq1,
                         const double* h_points1, const double* h_weights1, int use_fabs, int clamp, double* h_G,
                         double* h_detJ);

/* ---- operators -----------------------------------------------------------*/
typedef enum {
  WF_OP_STIFFNESS = 0,   /* StiffnessOperator      common/operators.hpp:137-201 */
  WF_OP_MASS_LUMPED = 1, /* MassOperatorCPU / SpectralMassOperator
                            common/operators.hpp:44-109, cuda/spectral_mass.hpp:24-100 */
  WF_OP_MASS_DENSE = 2   /* MassOperator (Phi^T D Phi) common/cuda/mass.hpp:18-107    */
} wf_op_kind;

typedef enum {
  WF_FLAG_NONE = 0,
  WF_FLAG_NO_FABS = 1,     /* detJ keeps its sign (spectral_mass.hpp:58-64)    */
  WF_FLAG_NO_CLAMP = 2,    /* skip the -1/0/1 clamp of G                        */
  WF_FLAG_TENSOR_X_SLOWEST = 8, /* the caller's tensor index -- the domain of h_perm (or of
                              h_dofmap when h_perm is NULL) and the point index of h_G /
                              h_detJ -- is l' = (i n + j) n + k with i along x, the order of
                              Basix' tensor-product factorisation, instead of the engine's
                              l = i + n (j + n k).  The 3x3 axes of G are reference axes
                              0, 1, 2 = x, y, z in both conventions.                   */
  WF_FLAG_MASS_ELEMENTWISE = 4 /* lumped mass with a dofmap: apply as the reference's
                              gather * detJ -> scatter-add per cell
                              (spectral_mass.hpp:84-89) instead of the pre-assembled
                              diagonal y += m .* x (m = M 1 built once at create)  */
} wf_flags;

/* Which kernel an operator runs (wf_op_info_t.kernel) ... */
typedef enum {
  WF_KERNEL_NONE = 0,
  WF_KERNEL_MARCH_BOX = 1,      /* k_stiffness_march: box mesh, implicit dofmap (stiffness_march.hip)       */
  WF_KERNEL_MARCH_IDX = 2,      /* k_march_idx: lattice columns of any dofmap (stiffness_march_idx.hip)     */
  WF_KERNEL_BATCH_UNIQUE = 3,   /* k_stiffness_generic_u / k_mass_lumped_u / k_mass_dense_col: batches of
                                   cells with a unique-dof tile -- what a mesh that does not tile gets     */
  WF_KERNEL_BOX_BLOCK = 4,      /* k_stiffness_box: one block of cells per workgroup, single pass           */
  WF_KERNEL_DIAGONAL = 5,       /* pre-assembled lumped mass, y += m .* x                                   */
  WF_KERNEL_MASS_DENSE_ANY = 6, /* k_mass_dense: any tensor rule (nq1 != P+1 allowed)                       */
  WF_KERNEL_DENSE_SIMPLEX = 7,  /* k_stiffness_dense: MFMA fp64, affine simplices                           */
  WF_KERNEL_ELEMENTWISE = 8     /* k_mass_lumped: one thread per element-local dof                          */
} wf_kernel_id;

/* ... and the explicit, thread-safe way to select one (tests, tuning runs).  Every field 0 =
 * the library's own choice; a NULL wf_tuning* means all defaults.  The library reads no
 * environment variable when it creates or applies an operator. */
typedef enum {
  WF_KERNEL_AUTO = 0,
  WF_KERNEL_FORCE_BATCH = 1,      /* skip the lattice-column plan: the batch kernels                        */
  WF_KERNEL_FORCE_BOX_BLOCK = 2,  /* box stiffness: the single-pass block kernel instead of marching        */
  WF_KERNEL_FORCE_MASS_ANY = 3,   /* dense mass: the any-rule kernel even for square tables                 */
  WF_KERNEL_FORCE_ELEMENTWISE = 4,/* batch kernels without the unique-dof tile (element-wise atomics)       */
  WF_KERNEL_FORCE_MARCH = 5       /* adopt the lattice-column plan whatever its fill (default: plans whose
                                     columns are mostly empty keep the batch kernel)                        */
} wf_kernel_hint;
typedef struct {
  int kernel;        /* wf_kernel_hint                                                              */
  int variant;       /* box marching kernel: compiled column cross-section + 1 (0 = default)        */
  int lz;            /* layers per z segment of the marching kernels (0 = chosen from the mesh)     */
  int lz0;           /* length of the first z segment under a z ghost plane (0 = 3)                 */
  int bx, by, bz;    /* cells per block of the single-pass box kernel (0 = default)                 */
  int keep_cell_order; /* batch kernels: do not sort the cells by their smallest dof                */
  int orient;        /* lattice plan: 0 = normalise cell orientations (default), 1 = require the
                        cells to agree as given                                                     */
} wf_tuning;

typedef struct {
  int kind;                    /* wf_op_kind                                    */
  int degree;                  /* P in 1..7, hexahedron, nd = (P+1)^3           */
  int ncells;                  /* local cells (operators.hpp:153)                */
  int ndofs;                   /* length of x and y: owned + ghost dofs          */
  const int32_t* h_dofmap;     /* [ncells][nd], element ordering                 */
  const int32_t* h_perm;       /* tensor -> element ordering (the reference's
                                  `perm`, operators.hpp:24); NULL = identity      */
  /* geometry: either precomputed arrays in the reference layout ...            */
  const double* h_G;           /* [ncells][nq][3][3] (stiffness) or NULL         */
  const double* h_detJ;        /* [ncells][nq] (mass) or NULL                    */
  /* ... or the mesh, in which case the geometry is computed on the device      */
  int nverts;
  const double* h_xverts;      /* [nverts][3]                                    */
  const int32_t* h_geom_dofmap;/* [ncells][8]                                    */
  double c0;                   /* speed of sound; stiffness carries -c0^2
                                  (operators.hpp:114-115; reference fixes 1500)   */
  int flags;                   /* wf_flags                                       */
  /* WF_OP_MASS_DENSE only: 1-D interpolation matrix phi1[nq1][P+1] (row-major)
   * of a tensor-product rule; h_detJ is then [ncells][nq1^3].                   */
  int nq1;
  const double* h_phi1;
  /* WF_OP_MASS_DENSE with the mesh instead of h_detJ: the 1-D rule (points, weights
   * [nq1]) at which det J * w is computed on the device (mass.hpp:35-39).        */
  const double* h_qpts1;
  const double* h_qwts1;
  const wf_tuning* tuning;     /* NULL = defaults                                */
} wf_op_desc;

/* Op(V, degree[, params]) constructors: operators.hpp:53,149; mass.hpp:20;
 * spectral_mass.hpp:26.  Copies every host array; owns all device state. */
int wf_op_create(const wf_op_desc* desc, wf_op** out);

/* Structured box mesh (mesh::create_box, demo/gpu_operator/main.cpp:60-63) with
 * this engine's lexicographic numbering: vertex (a,b,c) -> a + (nx+1)(b + (ny+1)c),
 * dof (I,J,K) -> I + NX*(J + NY*K), NX = P*nx+1 ...  The dofmap is implicit
 * (never read from memory) and the geometry is computed on the device from
 * h_xverts [(nx+1)(ny+1)(nz+1)][3]. */
int wf_op_create_box(int kind, int degree, int nx, int ny, int nz,
                     const double* h_xverts, double c0, int flags, wf_op** out);
int wf_op_create_box_tuned(int kind, int degree, int nx, int ny, int nz, const double* h_xverts, double c0, int flags,
                           const wf_tuning* tuning, wf_op** out);

/* Dense (non-tensor-product) stiffness operator on affine simplex cells: the
 * reference's skernel (common/operators.hpp:113-133) fed with arbitrary dense
 * tables, as its cell loop (operators.hpp:183-200) would be for a tetrahedral
 * space.  h_dphi is the table the reference keeps as `_dphi` (operators.hpp:178):
 * [3][nq][nd], already clamped by the caller like operators.hpp:27-29;
 * h_weights [nq]; cells are affine tetrahedra given by h_geom_dofmap [ncells][4].
 * G = (J^-1 |det J| w_q) J^-T with the -1/0/1 clamp is formed on the fly
 * (precomputation.hpp:95-107).  Compiled shapes: Lagrange P1..P4. */
typedef struct {
  int nd, nq;                   /* dofs and quadrature points per cell            */
  int ncells, ndofs;
  const int32_t* h_dofmap;      /* [ncells][nd]                                   */
  const double* h_dphi;         /* [3][nq][nd]                                    */
  const double* h_weights;      /* [nq]                                           */
  int nverts;
  const double* h_xverts;       /* [nverts][3]                                    */
  const int32_t* h_geom_dofmap; /* [ncells][4]                                    */
  double c0;
  int flags;                    /* WF_FLAG_NO_CLAMP                               */
} wf_dense_desc;
int wf_op_create_dense_simplex(const wf_dense_desc* desc, wf_op** out);

/* op(x, y) / op.apply(x, y): y += A x.  operators.hpp:183, mass.hpp:76,
 * spectral_mass.hpp:84. */
int wf_op_apply(wf_op* op, const double* d_x, double* d_y, void* stream);

/* Interior / interface split of a stiffness operator on a domain-decomposed mesh: lets the
 * caller overlap the ghost updates (VectorUpdater::update_fwd_begin/_end and update_rev,
 * demo/gpu_scatter_mpi/VectorUpdater.hpp:106-143,157-199) with the work that touches no ghost
 * dof.  The operator's work items (columns of cells, stiffness_march*.hip) are sorted into
 * INTERFACE (the item's dof tile contains a ghost position) and INTERIOR;
 * apply(INTERIOR) + apply(INTERFACE) == apply.
 *   wf_op_set_ghost_dofs : any marching operator (box or arbitrary dofmap), ghosts given as
 *                          positions in the local array -- the same list the updater gets
 *                          (wf_updater_desc.ghost_positions, any order, duplicates allowed);
 *   wf_op_set_ghost_faces: box operators, ghosts = the lower lattice plane of the marked axes
 *                          (the Cartesian partition of SURVEY 8e).
 * WF_ERR_UNSUPPORTED for operators that run a batch kernel (no work items to sort). */
typedef enum {
  WF_PART_ALL = 0,
  WF_PART_INTERIOR = 1,    /* every work item that touches no ghost dof                   */
  WF_PART_INTERFACE = 2,
  WF_PART_INTERIOR_A = 3,  /* INTERIOR split in two halves: A hides the forward halo,    */
  WF_PART_INTERIOR_B = 4   /* B the reverse (add) halo; A + B == INTERIOR                */
} wf_part;
int wf_op_set_ghost_faces(wf_op* op, int ghost_x0, int ghost_y0, int ghost_z0);
int wf_op_set_ghost_dofs(wf_op* op, const int32_t* h_ghost_positions, int32_t nghosts);
int wf_op_apply_part(wf_op* op, const double* d_x, double* d_y, int part, void* stream);

typedef struct {
  int kind, degree, num_cells, num_dofs_cell, num_quads, ndofs, structured;
  double flops;          /* reference model 4*ncells*nq*nd (mass.hpp:71)        */
  double alg_bytes;      /* algorithmic HBM bytes per apply          */
  size_t device_bytes;   /* device memory owned by the operator                  */
  int items_interior, items_interface; /* work items per part (wf_op_set_ghost_*)     */
  int kernel;            /* wf_kernel_id: the kernel wf_op_apply launches               */
  int plan_items, plan_patterns, plan_lz; /* lattice-column plan (WF_KERNEL_MARCH_IDX): work items,
                            distinct index tables, layers per item; 0 otherwise          */
  int plan_reoriented;   /* cells whose local axes the plan rotated / reflected to make them agree */
  double plan_fill;      /* cells / cell slots of the plan's columns                     */
} wf_op_info_t;
int wf_op_info(const wf_op* op, wf_op_info_t* info); /* num_quads()/num_cells()/... mass.hpp:68-71 */
int wf_op_destroy(wf_op* op);

/* ---- a8/a9: free kernels --------------------------------------------------
 * common/cuda/scatter.hpp:7-14, transform.hpp:7-8 (the reference passes a block
 * size; launch geometry is internal here). */
int wf_gather(int32_t N, const int32_t* d_indices, const double* d_in, double* d_out, void* stream);      /* out[i] = in[indices[i]]   */
int wf_scatter_add(int32_t N, const int32_t* d_indices, const double* d_in, double* d_out, void* stream); /* out[indices[i]] += in[i]  */
int wf_scatter_set(int32_t N, const int32_t* d_indices, const double* d_in, double* d_out, void* stream); /* out[indices[i]]  = in[i]  (VectorUpdater.hpp:141 unpack) */
int wf_transform1(int32_t N, const double* d_in, const double* d_detJ, double* d_out, void* stream);      /* out[i] = in[i]*detJ[i]    */

/* ---- a12: tall-skinny dense matmul (TSMM) ---------------------------------
 * out[cell][n] = sum_k in[cell][k] * phi[k][n], phi row-major [K][N] on the device.
 * Replaces the cublasDgemm pair of demo/gpu_tsmm/main.cpp:49-52 and the B / B^T
 * products of demo/gpu_operator/main.cpp:149-155 (fp64 MFMA 16x16x4).
 * layout 0: in[cell*K + k], out[cell*N + n]   (cell-major, demo/gpu_operator)
 * layout 1: in[k*ncells + cell], out[n*ncells + cell]   (the column-major arrays
 *           of demo/gpu_tsmm with lda = ldc = ncells).
 * Any K, N > 0: columns run in passes of 128, table rows in ranges of <= 128 that
 * accumulate onto out (so out is read as well as written when K > 128). */
int wf_tsmm(int layout, int64_t ncells, int K, int N, const double* d_in, const double* d_phi, double* d_out,
            void* stream);

/* ---- a15/a16: vector kernels of the RK4 loop ------------------------------
 * common/LinearGLL.hpp:17-33 (copy, axpy), :173 (fill), :182-191 (divide),
 * common/cuda/la.hpp:31-138. */
int wf_copy(int64_t n, const double* d_in, double* d_out, void* stream);
int wf_fill(int64_t n, double value, double* d_out, void* stream);
int wf_axpy(int64_t n, double alpha, const double* d_x, const double* d_y, double* d_r, void* stream); /* r = alpha*x + y */
int wf_scale(int64_t n, double alpha, double* d_x, void* stream);
int wf_pointwise_div(int64_t n, const double* d_b, const double* d_m, double* d_out, void* stream);    /* out = b / m */
int wf_pointwise_mult_add(int64_t n, const double* d_m, const double* d_x, double* d_y, void* stream); /* y += m .* x */
int wf_dot(int64_t n, const double* d_x, const double* d_y, double* d_result, void* stream);           /* la.hpp:87 inner_product; result on device */

/* Fused vector algebra between two stiffness applies of the RK4 loop
 * (common/LinearGLL.hpp:182-191, :260, :264-265 and the next stage's :250-254):
 *   kv = b/m; ku = vn; u = ku*bdt + u_read; v = kv*bdt + v_read; b = 0 (LinearGLL.hpp:173);
 *   if has_next: un = ku*adt_next + u0; vn_next = kv*adt_next + v0.
 * u_read/v_read may alias u/v; vn_next must not alias vn. */
int wf_rk4_stage(int64_t n, double bdt, double adt_next, int has_next, double* d_b, const double* d_m,
                 const double* d_vn, const double* d_u_read, const double* d_v_read, double* d_u, double* d_v,
                 const double* d_u0, const double* d_v0, double* d_un, double* d_vn_next, void* stream);

/* ---- a7: boundary operator (diagonal form of forms.ufl:19-24) -------------
 * b[idx1[i]] += s1 * m1[i];  b[idx2[i]] += s2 * m2[i] * v[idx2[i]].
 * LinearGLL.hpp:175 with s1 = c0^2 g(t), s2 = -c0. */
int wf_boundary_apply(int32_t n1, const int32_t* d_idx1, const double* d_m1, double s1,
                      int32_t n2, const int32_t* d_idx2, const double* d_m2, double s2,
                      const double* d_v, double* d_b, void* stream);

/* The boundary term folded into the fused stage: a plan of the two boundary dof sets (host arrays, as
 * wf_boundary_apply takes them on the device) -- a bitmap over the n dofs, a running count per 64 dofs and
 * the facet masses of the union set in dof order.  wf_rk4_stage_bc is wf_rk4_stage that, instead of zeroing
 * b, leaves the NEXT right-hand side's boundary term in it:
 *   b[i] = s1_next m1[i] + s2 m2[i] v'[i],  v' = vn_next (has_next) or the updated v (last stage),
 * so the separate wf_boundary_apply launch per stage disappears (LinearGLL.hpp:173-175; the stiffness apply
 * then accumulates onto it).  wf_boundary_apply_plan is the plain launch (first right-hand side of a run). */
typedef struct wf_boundary wf_boundary;
int wf_boundary_create(int64_t n, int32_t n1, const int32_t* h_idx1, const double* h_m1, int32_t n2, const int32_t* h_idx2,
                       const double* h_m2, wf_boundary** out);
int wf_boundary_destroy(wf_boundary* bc);
int wf_boundary_apply_plan(const wf_boundary* bc, double s1, double s2, const double* d_v, double* d_b, void* stream);
int wf_rk4_stage_bc(int64_t n, double bdt, double adt_next, int has_next, double* d_b, const double* d_m,
                    const double* d_vn, const double* d_u_read, const double* d_v_read, double* d_u, double* d_v,
                    const double* d_u0, const double* d_v0, double* d_un, double* d_vn_next, const wf_boundary* bc,
                    double s1_next, double s2, void* stream);

/* ---- a14: ghost exchange over RCCL ----------------------------------------
 * demo/gpu_scatter_mpi/VectorUpdater.hpp:21-230 (GPU pack + CUDA-aware MPI per
 * IndexMap neighbour) and la::Vector::scatter_fwd / scatter_rev(add) of
 * common/LinearGLL.hpp:110,127,164-176.  One process per GPU; the transport is a
 * grouped ncclSend/ncclRecv per neighbour (RCCL over xGMI) enqueued on a HIP
 * stream -- no host synchronisation anywhere.  librccl is bound at run time.
 *
 * Communicator: rank 0 calls wf_comm_unique_id, the launcher distributes the 128
 * bytes (MPI_Bcast, torch.distributed, or the file rendezvous below), every rank
 * calls wf_comm_create after wf_set_device. */
#define WF_COMM_ID_BYTES 128
typedef struct wf_comm wf_comm;
typedef struct wf_updater wf_updater;
typedef enum { WF_SUM = 0, WF_MAX = 1 } wf_reduce_op;

int wf_comm_unique_id(char* id /* [WF_COMM_ID_BYTES] */);
int wf_comm_create(const char* id, int rank, int nranks, wf_comm** out);
/* rank 0 publishes the id in `path` (unique per job), the others poll up to timeout_s */
int wf_comm_rendezvous_file(const char* path, int rank, double timeout_s, char* id /* [WF_COMM_ID_BYTES] */);
int wf_comm_create_from_file(const char* path, int rank, int nranks, double timeout_s, wf_comm** out);
int wf_comm_info(const wf_comm* comm, int* rank, int* nranks, int* rccl_version); /* outputs may be NULL */
/* MPI_Allreduce of demo/gpu_cg/CUDA/cg.hpp:21 on device scalars / arrays (in place allowed) */
int wf_comm_allreduce(wf_comm* comm, int op, int64_t count, const double* d_in, double* d_out, void* stream);
int wf_comm_barrier(wf_comm* comm, void* stream);   /* all ranks reached this point; synchronises `stream` */
int wf_comm_destroy(wf_comm* comm);

/* The IndexMap data VectorUpdater's constructor reads (VectorUpdater.hpp:31-98).
 * Segment i of send_indices [send_offsets[i], send_offsets[i+1]) lists the OWNED
 * local dofs that send_neighbors[i] holds as ghosts (scatter_fwd_indices);
 * segment i of ghost_positions lists the local positions of the ghosts owned by
 * recv_neighbors[i] in the order that neighbour sends them.  Positions index the
 * whole local array (a DOLFINx caller passes size_local + ghost index).  A rank may
 * be its own neighbour (periodic partition). */
typedef enum {
  WF_UPDATER_DEFAULT = 0,
  WF_UPDATER_INLINE = 1,  /* enqueue the exchange on the caller's stream instead of the
                             updater's communication stream (begin/end then do not overlap) */
  WF_UPDATER_CHAIN_ON_SIDE = 2 /* wf_op_apply_overlapped: halo chain on the updater's high-priority
                             stream and the interior on the caller's (default: the reverse)   */
} wf_updater_flags;
typedef struct {
  int ndofs;                          /* local array length: owned + ghosts               */
  int num_send_neighbors;
  const int* send_neighbors;          /* [num_send_neighbors] ranks                        */
  const int32_t* send_offsets;        /* [num_send_neighbors + 1], send_offsets[0] = 0     */
  const int32_t* send_indices;        /* [send_offsets[last]]                              */
  int num_recv_neighbors;
  const int* recv_neighbors;
  const int32_t* recv_offsets;        /* [num_recv_neighbors + 1]                          */
  const int32_t* ghost_positions;     /* [recv_offsets[last]]                              */
  int flags;                          /* wf_updater_flags                                  */
} wf_updater_desc;
int wf_updater_create(wf_comm* comm, const wf_updater_desc* desc, wf_updater** out);
/* update_fwd: owners -> ghosts (VectorUpdater.hpp:106-152).  begin packs on `stream`
 * and posts the exchange on the updater's own stream; work enqueued on `stream`
 * between begin and end overlaps with it; end makes `stream` wait and unpacks.
 * One exchange may be in flight per updater. */
int wf_updater_fwd_begin(wf_updater* u, const double* d_x, void* stream);
int wf_updater_fwd_end(wf_updater* u, double* d_x, void* stream);
int wf_updater_fwd(wf_updater* u, double* d_x, void* stream);
/* update_rev: ghosts -> owners, accumulating (VectorUpdater.hpp:157-208) */
int wf_updater_rev_begin(wf_updater* u, const double* d_x, void* stream);
int wf_updater_rev_end(wf_updater* u, double* d_x, void* stream);
int wf_updater_rev(wf_updater* u, double* d_x, void* stream);
int wf_updater_info(const wf_updater* u, int* num_send, int* num_recv, int* num_send_neighbors, int* num_recv_neighbors);
int wf_updater_destroy(wf_updater* u);

/* y += A x on a domain-decomposed mesh (LinearGLL.hpp:164-176 = scatter_fwd(x); apply;
 * scatter_rev(y)) with both halo directions hidden: update_fwd(x) -> apply(INTERFACE) ->
 * update_rev(y) runs on `stream` (the longer chain, so `stream` continues behind the reverse
 * unpack with no cross-stream wait on its critical path), apply(INTERIOR) beside it on a
 * low-priority stream of the updater; `stream` continues after both.
 * Needs wf_op_set_ghost_dofs / wf_op_set_ghost_faces. */
int wf_op_apply_overlapped(wf_op* op, wf_updater* u, double* d_x, double* d_y, void* stream);

/* ---- 8f: matrix-free conjugate gradients (BP1) -----------------------------
 * device::cg(x, b, matvec, kmax, rtol) of demo/gpu_cg/CUDA/cg.hpp:38-121: solves
 * A x = b for a symmetric positive definite operator, x holding the initial guess;
 * stops when ||r||^2 / ||r0||^2 < rtol^2 (cg.hpp:103) or after kmax iterations.
 * A is an operator handle or a callback with the library's accumulate semantics
 * (y += A v; wf_cg zeroes y first).  On a partitioned mesh pass the updater and
 * its communicator: the halo of the direction is updated before, the product
 * accumulated to the owners after every matvec, and the reductions run over owned
 * entries with one scalar all-reduce each (the MPI_Allreduce of cg.hpp:15-24).
 * The textbook algorithm is implemented, not the reference loop's arithmetic
 * (update_rev(p), nrm2 used as a squared norm, axpy(1,p,r): cg.hpp:84,59,117). */
typedef int (*wf_matvec_fn)(void* user, const double* d_v, double* d_y, void* stream);
typedef struct {
  int64_t n;              /* local vector length (owned + ghost)                  */
  wf_op* op;              /* A; used when matvec is NULL                           */
  wf_matvec_fn matvec;    /* or a callback, y += A v                               */
  void* user;
  wf_updater* updater;    /* NULL on one rank                                       */
  wf_comm* comm;          /* NULL on one rank                                       */
  int kmax;               /* reference default 50                                   */
  double rtol;            /* reference default 1e-8                                 */
} wf_cg_desc;
int wf_cg(const wf_cg_desc* desc, double* d_x, const double* d_b, int* iterations, double* rel_residual, void* stream);

/* ---- 8f: mesh / facet-tag input (XDMF + HDF5) -------------------------------
 * demo/cpu_planar3d/main.cpp:39-45: io::XDMFFile("mesh.xdmf").read_mesh(element, ghost_mode,
 * "planar3d") and read_meshtags(mesh, "planar3d_boundaries").  Host-only; libhdf5 is bound
 * at run time.  Hexahedral meshes; vertex orders are converted to the engine's tensor order
 * (cells: v = a + 2b + 4c, facets: v = a + 2b).  Facet tags come back as the four vertices of
 * every tagged facet plus its value. */
typedef struct wf_mesh_file wf_mesh_file;
int wf_mesh_open(const char* xdmf_path, const char* grid_name, wf_mesh_file** out);
int wf_mesh_sizes(wf_mesh_file* m, int64_t* nverts, int64_t* ncells);
int wf_mesh_read(wf_mesh_file* m, double* h_xverts /* [nverts][3] */, int32_t* h_cells /* [ncells][8] */);
int wf_mesh_tags_size(wf_mesh_file* m, const char* tags_name, int64_t* nfacets);
int wf_mesh_read_tags(wf_mesh_file* m, const char* tags_name, int32_t* h_facet_verts /* [nfacets][4] */,
                      int32_t* h_values /* [nfacets] */);
int wf_mesh_close(wf_mesh_file* m);
/* writer in the same layout (tags_name may be NULL); <name>.xdmf + <name>.h5 */
int wf_mesh_write(const char* xdmf_path, const char* grid_name, int64_t nverts, const double* h_xverts, int64_t ncells,
                  const int32_t* h_cells, const char* tags_name, int64_t nfacets, const int32_t* h_facet_verts,
                  const int32_t* h_values);

/* ---- range markers (roctx) -----------------------------------------------------
 * The nvtxMarkA / cudaProfilerStart bracketing of demo/gpu_scatter_mpi/main.cpp:89,101-121.
 * wf_markers_enable(1) binds librocprofiler-sdk-roctx at run time; from then on wf_op_apply*,
 * wf_updater_*, wf_rk4_stage and wf_cg push / pop a named range on the calling host thread
 * (rocprofv3 --marker-trace).  Off by default; host code can add its own ranges. */
int wf_markers_enable(int on);
int wf_marker_push(const char* name);
int wf_marker_pop(void);
int wf_marker_mark(const char* name);

/* ---- 8f: what DOLFINx derives from the mesh for the CPU demo (host only) ------------------
 * demo/cpu_planar3d/main.cpp:39-66, common/LinearGLL.hpp:113-115.  Cells and facets in the
 * engine's tensor vertex order (wf_mesh_read / wf_mesh_read_tags deliver it); cells may have ANY
 * local orientation.
 * wf_fs_build: fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped)): dofs are
 *   identified topologically (vertex / edge / face / interior entity + canonical position) and
 *   numbered in lexicographic (z, y, x) order of their coordinates.  h_dofmap [ncells][(P+1)^3],
 *   tensor order of each cell's own frame; h_dof_coords [coords_capacity >= *ndofs][3] may be NULL.
 * wf_fs_locate_facets: (cell, axis, side) of facets given by their four vertices (meshtags).
 * wf_fs_facet_mass: the tagged boundary dof set and its collocated facet masses
 *   m[i] = sum_facets w_q |dx/ds x dx/dt| (diagonal GLL form of forms.ufl:19-24), ascending dofs;
 *   h_idx / h_mass need capacity nfacets (P+1)^2.
 * wf_fs_min_cell_diameter: min over cells of mesh::h (largest vertex distance), main.cpp:48-57. */
int wf_fs_build(int degree, int64_t nverts, const double* h_xverts, int64_t ncells, const int32_t* h_cells,
                int64_t* ndofs, int32_t* h_dofmap, double* h_dof_coords, int64_t coords_capacity);
int wf_fs_locate_facets(int64_t ncells, const int32_t* h_cells, int64_t nfacets, const int32_t* h_facet_verts,
                        int32_t* h_cell, int32_t* h_axis, int32_t* h_side);
int wf_fs_facet_mass(int degree, int64_t nverts, const double* h_xverts, int64_t ncells, const int32_t* h_cells,
                     const int32_t* h_dofmap, int64_t nfacets, const int32_t* h_cell, const int32_t* h_axis,
                     const int32_t* h_side, int64_t* nout, int32_t* h_idx, double* h_mass);
int wf_fs_min_cell_diameter(int64_t nverts, const double* h_xverts, int64_t ncells, const int32_t* h_cells, double* hmin);

#ifdef __cplusplus
}
#endif
#endif /* WAVEHIP_H */
