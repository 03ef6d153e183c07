// wavehip.hpp -- header-only C++17 wrappers over the C ABI (wavehip.h) that keep
// the reference's class names, constructor arguments and apply semantics, so
// that reference-side code (common/LinearGLL.hpp, the demo mains) can switch by
// changing an include.  Exceptions: the C ABI never throws; these wrappers
// re-throw std::runtime_error exactly where the reference does
// (common/cuda/array.hpp:15-17,33-35; mass.hpp:91-92; utils.hpp:30-34).
//
// The operator classes are templated on a Vector concept satisfied by
// dolfinx::la::Vector<T, Alloc> (x.array().data(), y.mutable_array().data());
// the vectors must live in device memory (hipMalloc; see wavehip::allocator).
// The function-space argument of the reference is replaced by a plain
// `wavehip::Space` aggregate; INTEGRATION.md shows how to fill it from a
// dolfinx::fem::FunctionSpace.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "wavehip.h"

namespace wavehip {

inline void check(int rc)
{
  if (rc != WF_OK) throw std::runtime_error(std::string("wavehip: ") + wf_last_error());
}

// ---- utils::set_device (common/cuda/utils.hpp:22-38) ----------------------
inline int set_device(int rank)
{
  check(wf_set_device(rank));
  return rank;
}

// ---- cuda::array<T> (common/cuda/array.hpp:8-51) ---------------------------
template <class T>
class array {
public:
  array() = default;
  explicit array(std::size_t size) : _size(size) { check(wf_malloc(reinterpret_cast<void**>(&_data), size * sizeof(T))); }
  array(const array&) = delete;
  array& operator=(const array&) = delete;
  array(array&& o) noexcept : _data(o._data), _size(o._size) { o._data = nullptr; o._size = 0; }
  ~array() { wf_free(_data); }
  std::size_t size() const { return _size; }
  const T* data() const { return _data; }
  T* data() { return _data; }
  template <class Container>
  void set(const Container& source)
  {
    if (source.size() != _size) throw std::runtime_error("wavehip::array::set: size mismatch");
    check(wf_memcpy_h2d(_data, source.data(), _size * sizeof(T)));
  }
  std::vector<T> copy_to_host() const
  {
    std::vector<T> h(_size);
    check(wf_memcpy_d2h(h.data(), _data, _size * sizeof(T)));
    return h;
  }

private:
  T* _data = nullptr;
  std::size_t _size = 0;
};

// ---- CUDA::allocator<T> (common/cuda/allocator.hpp:9-40) -------------------
// std-allocator over device memory for dolfinx::la::Vector<T, Alloc>.  The
// reference uses managed memory; MI355X-native code uses explicit device buffers,
// so host code must not dereference vectors allocated with it.
template <class T>
struct allocator {
  using value_type = T;
  allocator() = default;
  template <class U>
  allocator(const allocator<U>&) noexcept {}
  T* allocate(std::size_t n)
  {
    void* p = nullptr;
    check(wf_malloc(&p, n * sizeof(T)));
    return static_cast<T*>(p);
  }
  void deallocate(T* p, std::size_t) noexcept { wf_free(p); }
  template <class U>
  bool operator==(const allocator<U>&) const noexcept { return true; }
  template <class U>
  bool operator!=(const allocator<U>&) const noexcept { return false; }
};

// ---- free kernels (common/cuda/scatter.hpp:7-14, transform.hpp:7-8) --------
template <typename T>
void gather(std::int32_t N, const std::int32_t* indices, const T* in, T* out, int /*block_size*/ = 512, void* stream = nullptr)
{
  check(wf_gather(N, indices, in, out, stream));
}
template <typename T>
void scatter(std::int32_t N, const std::int32_t* indices, const T* in, T* out, int /*block_size*/ = 512, void* stream = nullptr)
{
  check(wf_scatter_add(N, indices, in, out, stream));
}
template <typename T>
void transform1(std::int32_t N, const T* in, const T* detJ, T* out, int /*block_size*/ = 512, void* stream = nullptr)
{
  check(wf_transform1(N, in, detJ, out, stream));
}

// What the operators need from a dolfinx::fem::FunctionSpace (host arrays).
struct Space {
  int degree = 0;
  std::int32_t ncells = 0;                  // mesh->topology().index_map(tdim)->size_local()
  std::int32_t ndofs = 0;                   // index_map->size_local() + num_ghosts()
  const std::int32_t* dofmap = nullptr;     // V->dofmap()->list().array().data(), [ncells][nd]
  const std::int32_t* perm = nullptr;       // element.get_tensor_product_representation()[0] perm, or nullptr
  int tensor_flags = 0;                     // WF_FLAG_TENSOR_X_SLOWEST when perm (or the dofmap, or handed-over
                                            // G / detJ) is in Basix' tensor order l' = (i n + j) n + k
  std::int32_t nverts = 0;
  const double* x = nullptr;                // mesh->geometry().x().data(), [nverts][3]
  const std::int32_t* geom_dofmap = nullptr;// mesh->geometry().dofmap().array().data(), [ncells][8]
};

namespace detail {
class OpBase {
public:
  OpBase() = default;
  OpBase(const OpBase&) = delete;
  OpBase& operator=(const OpBase&) = delete;
  virtual ~OpBase() { wf_op_destroy(_op); }
  std::size_t num_quads() const { return info().num_quads; }
  std::size_t num_cells() const { return info().num_cells; }
  std::size_t num_dofs() const { return info().num_dofs_cell; }
  double flops() const { return info().flops; }

  /// y += A x   (common/operators.hpp:183, common/cuda/mass.hpp:76)
  template <typename Vector>
  void apply(const Vector& x, Vector& y, void* stream = nullptr)
  {
    check(wf_op_apply(_op, x.array().data(), y.mutable_array().data(), stream));
  }
  template <typename Vector>
  void operator()(const Vector& x, Vector& y) { apply(x, y); }
  void apply(const double* d_x, double* d_y, void* stream = nullptr) { check(wf_op_apply(_op, d_x, d_y, stream)); }
  wf_op* handle() const { return _op; }

protected:
  wf_op_info_t info() const
  {
    wf_op_info_t i{};
    check(wf_op_info(_op, &i));
    return i;
  }
  static wf_op_desc base_desc(const Space& V, int kind, int degree)
  {
    wf_op_desc d{};
    d.kind = kind;
    d.degree = degree;
    d.ncells = V.ncells;
    d.ndofs = V.ndofs;
    d.h_dofmap = V.dofmap;
    d.h_perm = V.perm;
    d.flags = V.tensor_flags;
    d.nverts = V.nverts;
    d.h_xverts = V.x;
    d.h_geom_dofmap = V.geom_dofmap;
    return d;
  }
  wf_op* _op = nullptr;
};
}  // namespace detail

/// StiffnessOperator(V, bdegree, params) -- common/operators.hpp:137-201
template <typename T>
class StiffnessOperator : public detail::OpBase {
  static_assert(sizeof(T) == sizeof(double), "fp64 only");

public:
  StiffnessOperator(const Space& V, int bdegree, std::map<std::string, double>& params)
  {
    wf_op_desc d = base_desc(V, WF_OP_STIFFNESS, bdegree);
    auto it = params.find("c0");
    d.c0 = it == params.end() ? 1500.0 : it->second;   // operators.hpp:114
    check(wf_op_create(&d, &_op));
  }
};

/// MassOperatorCPU(V, bdegree) -- common/operators.hpp:44-109 (the `MassOperator`
/// that common/LinearGLL.hpp:63,105 instantiates)
template <typename T>
class MassOperatorLumped : public detail::OpBase {
public:
  MassOperatorLumped(const Space& V, int bdegree, int flags = WF_FLAG_NONE)
  {
    wf_op_desc d = base_desc(V, WF_OP_MASS_LUMPED, bdegree);
    d.flags |= flags;
    check(wf_op_create(&d, &_op));
  }
};

/// SpectralMassOperator(V, bdegree) -- common/cuda/spectral_mass.hpp:24-100
template <typename T>
class SpectralMassOperator : public MassOperatorLumped<T> {
public:
  SpectralMassOperator(const Space& V, int bdegree) : MassOperatorLumped<T>(check_degree(V, bdegree), bdegree, WF_FLAG_NO_FABS) {}

private:
  static const Space& check_degree(const Space& V, int bdegree)
  {
    if (bdegree < 2 || bdegree > 7) throw std::runtime_error("SpectralMassOperator: degree must be 2..7");
    return V;
  }
};

/// MassOperator(V, element, quad_type, qd) -- common/cuda/mass.hpp:18-107.
/// phi1: 1-D interpolation matrix [nq1][degree+1]; detJ: [ncells][nq1^3] (host).
template <typename T>
class MassOperator : public detail::OpBase {
public:
  MassOperator(const Space& V, int degree, int nq1, const double* phi1, const double* detJ)
  {
    wf_op_desc d = base_desc(V, WF_OP_MASS_DENSE, degree);
    d.nq1 = nq1;
    d.h_phi1 = phi1;
    d.h_detJ = detJ;
    check(wf_op_create(&d, &_op));
  }
  /// The reference's argument list (mass.hpp:20-21): element = degree + Lagrange variant
  /// (WF_VARIANT_GLL_WARPED / WF_VARIANT_EQUISPACED), quad_type (WF_QUAD_GLL /
  /// WF_QUAD_GAUSS_JACOBI) and quadrature degree qd.  Builds the 1-D table (tabulate_1d,
  /// precompute.hpp:179-189) on the host and det J * w at the rule's points on the device.
  MassOperator(const Space& V, int degree, int variant, int quad_type, int qd)
  {
    int nq1 = 0;
    double pts[WF_MAX_QUAD_POINTS], wts[WF_MAX_QUAD_POINTS];
    check(wf_quadrature_1d(quad_type, qd, &nq1, pts, wts));
    std::vector<double> phi1((std::size_t)nq1 * (degree + 1));
    check(wf_tabulate_1d(degree, variant, nq1, pts, 0, phi1.data()));
    wf_op_desc d = base_desc(V, WF_OP_MASS_DENSE, degree);
    d.nq1 = nq1;
    d.h_phi1 = phi1.data();
    d.h_qpts1 = pts;
    d.h_qwts1 = wts;
    d.flags |= WF_FLAG_NO_FABS;   // mass.hpp:35-39: det J * w keeps its sign (compute_jacobian_determinant)
    check(wf_op_create(&d, &_op));
  }
};

/// Structured box operators (mesh::create_box with this engine's numbering).
template <typename T>
class BoxStiffnessOperator : public detail::OpBase {
public:
  BoxStiffnessOperator(int degree, int nx, int ny, int nz, const double* xverts, double c0)
  {
    check(wf_op_create_box(WF_OP_STIFFNESS, degree, nx, ny, nz, xverts, c0, WF_FLAG_NONE, &_op));
  }
};

// ---- communicator + VectorUpdater (demo/gpu_scatter_mpi/VectorUpdater.hpp) ----
/// RCCL communicator behind the C ABI.  The reference's launcher is mpirun; a
/// launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torchrun --no-python,
/// a shell loop, SLURM) works with from_env(), which meets through a file.
class Comm {
public:
  Comm(const char* id, int rank, int nranks) { check(wf_comm_create(id, rank, nranks, &_c)); }
  Comm(const std::string& rendezvous_file, int rank, int nranks, double timeout_s = 120.0)
  {
    check(wf_comm_create_from_file(rendezvous_file.c_str(), rank, nranks, timeout_s, &_c));
  }
  Comm(const Comm&) = delete;
  Comm& operator=(const Comm&) = delete;
  ~Comm() { wf_comm_destroy(_c); }

  static int env_int(const char* name, int dflt)
  {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
  }
  /// One rank per process: device LOCAL_RANK, rendezvous file $WF_COMM_FILE or
  /// /tmp/wavehip_comm_$MASTER_PORT.  Calls set_device.
  static std::unique_ptr<Comm> from_env()
  {
    const int rank = env_int("RANK", 0), size = env_int("WORLD_SIZE", 1);
    set_device(env_int("LOCAL_RANK", rank));
    const char* f = std::getenv("WF_COMM_FILE");
    std::string path = f ? f : std::string("/tmp/wavehip_comm_") + std::to_string(env_int("MASTER_PORT", 29500));
    return std::make_unique<Comm>(path, rank, size);
  }
  int rank() const
  {
    int r = 0;
    check(wf_comm_info(_c, &r, nullptr, nullptr));
    return r;
  }
  int size() const
  {
    int n = 1;
    check(wf_comm_info(_c, nullptr, &n, nullptr));
    return n;
  }
  /// MPI_Allreduce on device data (demo/gpu_cg/CUDA/cg.hpp:21)
  void allreduce(const double* d_in, double* d_out, std::int64_t count, int op = WF_SUM, void* stream = nullptr)
  {
    check(wf_comm_allreduce(_c, op, count, d_in, d_out, stream));
  }
  void barrier(void* stream = nullptr) { check(wf_comm_barrier(_c, stream)); }
  wf_comm* handle() const { return _c; }

private:
  wf_comm* _c = nullptr;
};

/// The IndexMap data the reference's VectorUpdater constructor extracts
/// (VectorUpdater.hpp:31-98): neighbours, displacements, scatter_fwd_indices and
/// ghost positions.  A DOLFINx caller fills it from
/// index_map->scatter_fwd_indices() / scatter_fwd_ghost_positions().
struct GhostLists {
  std::int32_t ndofs = 0;                       // size_local + num_ghosts
  std::vector<int> send_neighbors, recv_neighbors;
  std::vector<std::int32_t> send_offsets{0}, recv_offsets{0};
  std::vector<std::int32_t> send_indices;       // owned local dofs, per send neighbour
  std::vector<std::int32_t> ghost_positions;    // local positions of the ghosts, per recv neighbour
  bool empty() const { return send_neighbors.empty() && recv_neighbors.empty(); }
};

/// VectorUpdater<T, AllocatorT> -- demo/gpu_scatter_mpi/VectorUpdater.hpp:21-230.
/// Same method names; vectors are anything with array()/mutable_array() or raw
/// device pointers.  All calls are stream-ordered (no host synchronisation).
template <typename T>
class VectorUpdater {
  static_assert(sizeof(T) == sizeof(double), "fp64 only");

public:
  VectorUpdater(Comm* comm, const GhostLists& g, int flags = WF_UPDATER_DEFAULT)
  {
    wf_updater_desc d{};
    d.ndofs = g.ndofs;
    d.num_send_neighbors = (int)g.send_neighbors.size();
    d.send_neighbors = g.send_neighbors.data();
    d.send_offsets = g.send_offsets.data();
    d.send_indices = g.send_indices.data();
    d.num_recv_neighbors = (int)g.recv_neighbors.size();
    d.recv_neighbors = g.recv_neighbors.data();
    d.recv_offsets = g.recv_offsets.data();
    d.ghost_positions = g.ghost_positions.data();
    d.flags = flags;
    check(wf_updater_create(comm ? comm->handle() : nullptr, &d, &_u));
  }
  VectorUpdater(const VectorUpdater&) = delete;
  VectorUpdater& operator=(const VectorUpdater&) = delete;
  ~VectorUpdater() { wf_updater_destroy(_u); }

  void update_fwd_begin(const T* x, void* stream = nullptr) { check(wf_updater_fwd_begin(_u, x, stream)); }
  void update_fwd_begin(T* x, void* stream = nullptr) { update_fwd_begin(static_cast<const T*>(x), stream); }
  void update_rev_begin(T* x, void* stream = nullptr) { update_rev_begin(static_cast<const T*>(x), stream); }
  void update_fwd_end(T* x, void* stream = nullptr) { check(wf_updater_fwd_end(_u, x, stream)); }
  void update_fwd(T* x, void* stream = nullptr) { check(wf_updater_fwd(_u, x, stream)); }
  void update_rev_begin(const T* x, void* stream = nullptr) { check(wf_updater_rev_begin(_u, x, stream)); }
  void update_rev_end(T* x, void* stream = nullptr) { check(wf_updater_rev_end(_u, x, stream)); }
  void update_rev(T* x, void* stream = nullptr) { check(wf_updater_rev(_u, x, stream)); }
  template <typename Vector>
  void update_fwd_begin(const Vector& x) { update_fwd_begin(static_cast<const T*>(x.array().data())); }
  template <typename Vector>
  void update_fwd_end(Vector& x) { update_fwd_end(static_cast<T*>(x.mutable_array().data())); }
  template <typename Vector>
  void update_fwd(Vector& x) { update_fwd(static_cast<T*>(x.mutable_array().data())); }
  template <typename Vector>
  void update_rev_begin(const Vector& x) { update_rev_begin(static_cast<const T*>(x.array().data())); }
  template <typename Vector>
  void update_rev_end(Vector& x) { update_rev_end(static_cast<T*>(x.mutable_array().data())); }
  template <typename Vector>
  void update_rev(Vector& x) { update_rev(static_cast<T*>(x.mutable_array().data())); }
  wf_updater* handle() const { return _u; }

private:
  wf_updater* _u = nullptr;
};

// ---- device::cg (demo/gpu_cg/CUDA/cg.hpp:38-121) -------------------------------
namespace device {
/// Solve A x = b with the conjugate gradient method; returns the iteration count.
/// matvec(p, y) must ACCUMULATE y += A p on raw device pointers (cg zeroes y first),
/// e.g. [&](const double* p, double* y, void* s) { op.apply(p, y, s); }.  x holds the
/// initial guess.  On a partitioned mesh pass the VectorUpdater and its Comm.
/// Textbook CG with the reference's stopping rule (cg.hpp:103); see wf_cg.
template <typename T>
int cg(T* x, const T* b, std::int64_t n, std::function<void(const T*, T*, void*)> matvec, int kmax = 50,
       double rtol = 1e-8, VectorUpdater<T>* updater = nullptr, Comm* comm = nullptr, double* rel_residual = nullptr)
{
  static_assert(sizeof(T) == sizeof(double), "fp64 only");
  struct Ctx {
    std::function<void(const T*, T*, void*)>* f;
    std::string what;
  } ctx{&matvec, {}};
  wf_cg_desc d{};
  d.n = n;
  d.user = &ctx;
  d.matvec = [](void* user, const double* v, double* y, void* stream) -> int {
    auto* c = static_cast<Ctx*>(user);
    try {
      (*c->f)(v, y, stream);
      return 0;
    } catch (const std::exception& e) {   // exceptions must not cross the C boundary
      c->what = e.what();
      return -1;
    }
  };
  d.updater = updater ? updater->handle() : nullptr;
  d.comm = comm ? comm->handle() : nullptr;
  d.kmax = kmax;
  d.rtol = rtol;
  int its = 0;
  const int rc = wf_cg(&d, x, b, &its, rel_residual, nullptr);
  if (rc != WF_OK && !ctx.what.empty()) throw std::runtime_error("wavehip::device::cg: matvec threw: " + ctx.what);
  check(rc);
  return its;
}
/// Vector-concept overload with the reference's argument order.
template <typename Vector>
int cg(Vector& x, const Vector& b, std::function<void(const double*, double*, void*)> matvec, int kmax = 50,
       double rtol = 1e-8)
{
  return cg<double>(x.mutable_array().data(), b.array().data(), (std::int64_t)x.array().size(), std::move(matvec), kmax, rtol);
}
}  // namespace device

// ---- linalg:: (common/cuda/la.hpp:31-138) and kernels:: (LinearGLL.hpp:15-35)
namespace linalg {
template <typename Vector>
void copy(const Vector& x, Vector& y, void* stream = nullptr)
{
  check(wf_copy((std::int64_t)x.array().size(), x.array().data(), y.mutable_array().data(), stream));
}
template <typename Scalar, typename Vector>
void axpy(Scalar alpha, const Vector& x, Vector& y, void* stream = nullptr)   // y = alpha x + y over size_local
{
  check(wf_axpy((std::int64_t)x.map()->size_local(), (double)alpha, x.array().data(), y.array().data(),
                y.mutable_array().data(), stream));
}
template <typename Scalar, typename Vector>
void scale(Scalar alpha, Vector& x, void* stream = nullptr)
{
  check(wf_scale((std::int64_t)x.map()->size_local(), (double)alpha, x.mutable_array().data(), stream));
}
}  // namespace linalg

}  // namespace wavehip
