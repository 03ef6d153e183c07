// Ghost exchange behind the C ABI: RCCL communicator + VectorUpdater
// (demo/gpu_scatter_mpi/VectorUpdater.hpp:21-230).
//
// The reference packs on the GPU and posts one CUDA-aware MPI_Irecv/MPI_Send per
// IndexMap neighbour.  Here the transport is RCCL over xGMI: one grouped
// ncclSend/ncclRecv per neighbour (<= 7 peers on the 2x2x2 partition = one per
// xGMI link), enqueued on a HIP stream; pack = k_gather, unpack fwd = indexed
// store, unpack rev = indexed atomic add.  Nothing synchronises with the host.
//
// librccl is bound at run time (dlopen) so that libwavehip has no hard
// dependency on it and so that, inside a PyTorch process, the communicator is
// created in the RCCL build torch has already loaded (same SONAME librccl.so.1)
// instead of starting a second copy.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include "common.h"

namespace {

// the slice of the RCCL ABI used here (rccl.h of ROCm 7.x; values are part of the stable NCCL ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
enum { kNcclSuccess = 0, kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2 };

struct RcclApi {
  void* handle = nullptr;
  std::string path;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

RcclApi* g_rccl = nullptr;

template <typename F>
bool bind(void* h, const char* name, F* fn)
{
  *fn = reinterpret_cast<F>(dlsym(h, name));
  return *fn != nullptr;
}

// Loads librccl once.  Search order: $WF_RCCL_LIB, the SONAME (resolves to a copy
// already mapped into the process, e.g. PyTorch's), the ROCm install.
RcclApi* rccl()
{
  if (g_rccl) return g_rccl;
  std::vector<std::string> cands;
  if (const char* e = std::getenv("WF_RCCL_LIB")) cands.push_back(e);
  cands.push_back("librccl.so.1");
  cands.push_back("librccl.so");
  cands.push_back("/opt/rocm/lib/librccl.so.1");
  std::string tried;
  for (const auto& c : cands) {
    void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      tried += c + " (" + dlerror() + "); ";
      continue;
    }
    auto api = std::make_unique<RcclApi>();
    api->handle = h;
    api->path = c;
    bool ok = bind(h, "ncclGetVersion", &api->GetVersion) && bind(h, "ncclGetUniqueId", &api->GetUniqueId)
              && bind(h, "ncclCommInitRank", &api->CommInitRank) && bind(h, "ncclCommDestroy", &api->CommDestroy)
              && bind(h, "ncclGroupStart", &api->GroupStart) && bind(h, "ncclGroupEnd", &api->GroupEnd)
              && bind(h, "ncclSend", &api->Send) && bind(h, "ncclRecv", &api->Recv)
              && bind(h, "ncclAllReduce", &api->AllReduce) && bind(h, "ncclGetErrorString", &api->GetErrorString);
    if (!ok) {
      tried += c + " (missing symbols); ";
      dlclose(h);
      continue;
    }
    g_rccl = api.release();
    return g_rccl;
  }
  wf::set_error("RCCL not available: " + tried);
  return nullptr;
}

#define WF_NCCL_CHECK(api, expr)                                                              \
  do {                                                                                        \
    int _r = (expr);                                                                          \
    if (_r != kNcclSuccess) {                                                                 \
      ::wf::set_error(std::string(#expr) + " failed: " + (api)->GetErrorString(_r));           \
      return WF_ERR_COMM;                                                                     \
    }                                                                                         \
  } while (0)

}  // namespace

struct wf_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  double* d_scratch = nullptr;   // one double for barrier / scalar reductions
};

struct wf_updater {
  wf_comm* comm = nullptr;
  int ndofs = 0;
  std::vector<int> send_nb, recv_nb;
  std::vector<int32_t> send_off, recv_off;   // displs_send_fwd / displs_recv_fwd (VectorUpdater.hpp:34-46)
  int32_t* d_indices = nullptr;              // scatter_fwd_indices            (VectorUpdater.hpp:49-52)
  int32_t* d_ghost_pos = nullptr;            // scatter_fwd_ghost_positions    (VectorUpdater.hpp:55-59)
  double* d_send_buffer = nullptr;           // (VectorUpdater.hpp:62-63)
  double* d_recv_buffer = nullptr;
  int32_t nsend = 0, nrecv = 0;
  int flags = 0;
  hipStream_t comm_stream = nullptr;         // the exchange runs here unless WF_UPDATER_INLINE
  hipEvent_t ev_packed = nullptr, ev_done = nullptr;
  // streams of wf_op_apply_overlapped: a low-priority stream for the interior while the halo chain runs on
  // the caller's stream (default), or a high-priority side stream for the chain (WF_UPDATER_CHAIN_ON_SIDE).
  // (A CU-masked interior stream that keeps a few CUs free for the small pack / RCCL / unpack kernels was
  // measured and dropped: 0.476 vs 0.346 ms per step, every event wait to or from the masked queue 40-60 us.)
  hipStream_t side_stream = nullptr;
  hipStream_t low_stream = nullptr;
  hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_interior = nullptr;
};

namespace {

void free_updater(wf_updater* u)
{
  if (!u) return;
  (void)hipFree(u->d_indices);
  (void)hipFree(u->d_ghost_pos);
  (void)hipFree(u->d_send_buffer);
  (void)hipFree(u->d_recv_buffer);
  if (u->ev_packed) (void)hipEventDestroy(u->ev_packed);
  if (u->ev_done) (void)hipEventDestroy(u->ev_done);
  if (u->ev_main) (void)hipEventDestroy(u->ev_main);
  if (u->ev_side) (void)hipEventDestroy(u->ev_side);
  if (u->ev_interior) (void)hipEventDestroy(u->ev_interior);
  if (u->low_stream) (void)hipStreamDestroy(u->low_stream);
  if (u->comm_stream) (void)hipStreamDestroy(u->comm_stream);
  if (u->side_stream) (void)hipStreamDestroy(u->side_stream);
  delete u;
}

// One grouped neighbour exchange: segment i of `sendbuf` goes to send_nb[i], segment i
// of `recvbuf` comes from recv_nb[i]  (VectorUpdater.hpp:113-130 / :170-188).
int exchange(wf_updater* u, const double* sendbuf, const std::vector<int32_t>& soff, const std::vector<int>& snb,
             double* recvbuf, const std::vector<int32_t>& roff, const std::vector<int>& rnb, hipStream_t s)
{
  if (snb.empty() && rnb.empty()) return WF_OK;
  RcclApi* api = rccl();
  if (!api) return WF_ERR_COMM;
  WF_NCCL_CHECK(api, api->GroupStart());
  for (size_t i = 0; i < rnb.size(); ++i) {
    const size_t cnt = (size_t)(roff[i + 1] - roff[i]);
    if (cnt) WF_NCCL_CHECK(api, api->Recv(recvbuf + roff[i], cnt, kNcclFloat64, rnb[i], u->comm->comm, s));
  }
  for (size_t i = 0; i < snb.size(); ++i) {
    const size_t cnt = (size_t)(soff[i + 1] - soff[i]);
    if (cnt) WF_NCCL_CHECK(api, api->Send(sendbuf + soff[i], cnt, kNcclFloat64, snb[i], u->comm->comm, s));
  }
  WF_NCCL_CHECK(api, api->GroupEnd());
  return WF_OK;
}

// pack on `user`, exchange on the updater's stream (or inline on `user`)
int begin(wf_updater* u, const int32_t* d_pack_idx, int32_t npack, const double* d_x, double* packbuf,
          const std::vector<int32_t>& soff, const std::vector<int>& snb, double* recvbuf,
          const std::vector<int32_t>& roff, const std::vector<int>& rnb, hipStream_t user, bool inl)
{
  int rc = wf_gather(npack, d_pack_idx, d_x, packbuf, user);
  if (rc != WF_OK) return rc;
  if (inl) return exchange(u, packbuf, soff, snb, recvbuf, roff, rnb, user);
  WF_HIP_CHECK(hipEventRecord(u->ev_packed, user));
  WF_HIP_CHECK(hipStreamWaitEvent(u->comm_stream, u->ev_packed, 0));
  rc = exchange(u, packbuf, soff, snb, recvbuf, roff, rnb, u->comm_stream);
  if (rc != WF_OK) return rc;
  WF_HIP_CHECK(hipEventRecord(u->ev_done, u->comm_stream));
  return WF_OK;
}

int wait_exchange(wf_updater* u, hipStream_t user, bool inl)
{
  if (!inl) WF_HIP_CHECK(hipStreamWaitEvent(user, u->ev_done, 0));
  return WF_OK;
}

// VectorUpdater.hpp:106-131: pack the owned values the neighbours hold as ghosts, post the exchange
int fwd_begin(wf_updater* u, const double* d_x, hipStream_t s, bool inl)
{
  wf::MarkerScope mk("update_fwd_begin");
  return begin(u, u->d_indices, u->nsend, d_x, u->d_send_buffer, u->send_off, u->send_nb, u->d_recv_buffer, u->recv_off,
               u->recv_nb, s, inl);
}
// VectorUpdater.hpp:133-143: wait, copy into the ghost entries
int fwd_end(wf_updater* u, double* d_x, hipStream_t s, bool inl)
{
  wf::MarkerScope mk("update_fwd_end");
  int rc = wait_exchange(u, s, inl);
  if (rc != WF_OK) return rc;
  return wf_scatter_set(u->nrecv, u->d_ghost_pos, u->d_recv_buffer, d_x, s);
}
// VectorUpdater.hpp:157-189: the buffers swap roles: pack the ghost entries, send them to their owners
int rev_begin(wf_updater* u, const double* d_x, hipStream_t s, bool inl)
{
  wf::MarkerScope mk("update_rev_begin");
  return begin(u, u->d_ghost_pos, u->nrecv, d_x, u->d_recv_buffer, u->recv_off, u->recv_nb, u->d_send_buffer, u->send_off,
               u->send_nb, s, inl);
}
// VectorUpdater.hpp:191-199: wait, accumulate into the owned entries (atomic add, scatter.cu:43)
int rev_end(wf_updater* u, double* d_x, hipStream_t s, bool inl)
{
  wf::MarkerScope mk("update_rev_end");
  int rc = wait_exchange(u, s, inl);
  if (rc != WF_OK) return rc;
  return wf_scatter_add(u->nsend, u->d_indices, u->d_send_buffer, d_x, s);
}
inline bool is_inline(const wf_updater* u) { return (u->flags & WF_UPDATER_INLINE) != 0; }

}  // namespace

extern "C" {

int wf_comm_unique_id(char* id)
{
  WF_REQUIRE(id != nullptr, "wf_comm_unique_id: null output");
  RcclApi* api = rccl();
  if (!api) return WF_ERR_COMM;
  ncclUniqueId uid;
  WF_NCCL_CHECK(api, api->GetUniqueId(&uid));
  std::memcpy(id, uid.internal, WF_COMM_ID_BYTES);
  return WF_OK;
}

int wf_comm_create(const char* id, int rank, int nranks, wf_comm** out)
{
  WF_REQUIRE(id && out, "wf_comm_create: null argument");
  *out = nullptr;
  WF_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "wf_comm_create: bad rank / nranks");
  RcclApi* api = rccl();
  if (!api) return WF_ERR_COMM;
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, WF_COMM_ID_BYTES);
  auto c = std::make_unique<wf_comm>();
  c->rank = rank;
  c->nranks = nranks;
  WF_NCCL_CHECK(api, api->CommInitRank(&c->comm, nranks, uid, rank));
  WF_HIP_CHECK(hipMalloc((void**)&c->d_scratch, 2 * sizeof(double)));
  WF_HIP_CHECK(hipMemset(c->d_scratch, 0, 2 * sizeof(double)));
  *out = c.release();
  return WF_OK;
}

// Rendezvous through a file for launchers without MPI: rank 0 writes the id to
// `path` (atomically, via rename), the others poll for it.  `path` must be unique
// per job (e.g. contain MASTER_PORT).
int wf_comm_rendezvous_file(const char* path, int rank, double timeout_s, char* id)
{
  WF_REQUIRE(path && id && rank >= 0, "wf_comm_rendezvous_file: bad argument");
  if (rank == 0) {
    int rc = wf_comm_unique_id(id);
    if (rc != WF_OK) return rc;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, WF_COMM_ID_BYTES, f) != WF_COMM_ID_BYTES) {
      if (f) std::fclose(f);
      wf::set_error(std::string("wf_comm_rendezvous_file: cannot write ") + tmp);
      return WF_ERR_COMM;
    }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path) != 0) {
      wf::set_error(std::string("wf_comm_rendezvous_file: cannot publish ") + path);
      return WF_ERR_COMM;
    }
    return WF_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    struct stat st;
    if (stat(path, &st) == 0 && st.st_size == WF_COMM_ID_BYTES) {
      FILE* f = std::fopen(path, "rb");
      const bool ok = f && std::fread(id, 1, WF_COMM_ID_BYTES, f) == WF_COMM_ID_BYTES;
      if (f) std::fclose(f);
      if (ok) return WF_OK;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      wf::set_error(std::string("wf_comm_rendezvous_file: timed out waiting for ") + path);
      return WF_ERR_COMM;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

// rendezvous + ncclCommInitRank; rank 0 removes the file once the communicator is up
int wf_comm_create_from_file(const char* path, int rank, int nranks, double timeout_s, wf_comm** out)
{
  WF_REQUIRE(path && out, "wf_comm_create_from_file: null argument");
  *out = nullptr;
  char id[WF_COMM_ID_BYTES];
  int rc = wf_comm_rendezvous_file(path, rank, timeout_s, id);
  if (rc != WF_OK) return rc;
  rc = wf_comm_create(id, rank, nranks, out);
  if (rank == 0) (void)unlink(path);   // every rank has read it: CommInitRank returns only when all joined
  return rc;
}

int wf_comm_info(const wf_comm* comm, int* rank, int* nranks, int* rccl_version)
{
  WF_REQUIRE(comm != nullptr, "wf_comm_info: null handle");
  if (rank) *rank = comm->rank;
  if (nranks) *nranks = comm->nranks;
  if (rccl_version) {
    RcclApi* api = rccl();
    if (!api) return WF_ERR_COMM;
    WF_NCCL_CHECK(api, api->GetVersion(rccl_version));
  }
  return WF_OK;
}

int wf_comm_allreduce(wf_comm* comm, int op, int64_t count, const double* d_in, double* d_out, void* stream)
{
  WF_REQUIRE(comm && d_in && d_out && count >= 0, "wf_comm_allreduce: bad argument");
  WF_REQUIRE(op == WF_SUM || op == WF_MAX, "wf_comm_allreduce: op must be WF_SUM or WF_MAX");
  if (count == 0) return WF_OK;
  RcclApi* api = rccl();
  if (!api) return WF_ERR_COMM;
  WF_NCCL_CHECK(api, api->AllReduce(d_in, d_out, (size_t)count, kNcclFloat64, op == WF_SUM ? kNcclSum : kNcclMax,
                                    comm->comm, (hipStream_t)stream));
  return WF_OK;
}

int wf_comm_barrier(wf_comm* comm, void* stream)
{
  WF_REQUIRE(comm != nullptr, "wf_comm_barrier: null handle");
  int rc = wf_comm_allreduce(comm, WF_SUM, 1, comm->d_scratch, comm->d_scratch + 1, stream);
  if (rc != WF_OK) return rc;
  WF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return WF_OK;
}

int wf_comm_destroy(wf_comm* comm)
{
  if (!comm) return WF_OK;
  (void)hipFree(comm->d_scratch);
  if (comm->comm && g_rccl) (void)g_rccl->CommDestroy(comm->comm);
  delete comm;
  return WF_OK;
}

int wf_updater_create(wf_comm* comm, const wf_updater_desc* desc, wf_updater** out)
{
  WF_REQUIRE(desc && out, "wf_updater_create: null argument");
  *out = nullptr;
  WF_REQUIRE(desc->num_send_neighbors >= 0 && desc->num_recv_neighbors >= 0 && desc->ndofs >= 0,
             "wf_updater_create: negative size");
  WF_REQUIRE(comm || (desc->num_send_neighbors == 0 && desc->num_recv_neighbors == 0),
             "wf_updater_create: neighbours given without a communicator");
  std::unique_ptr<wf_updater, void (*)(wf_updater*)> u(new wf_updater, free_updater);
  u->comm = comm;
  u->ndofs = desc->ndofs;
  u->flags = desc->flags;
  const int ns = desc->num_send_neighbors, nr = desc->num_recv_neighbors;
  WF_REQUIRE(ns == 0 || (desc->send_neighbors && desc->send_offsets), "wf_updater_create: send lists missing");
  WF_REQUIRE(nr == 0 || (desc->recv_neighbors && desc->recv_offsets), "wf_updater_create: recv lists missing");
  u->send_off.assign(1, 0);
  u->recv_off.assign(1, 0);
  for (int i = 0; i < ns; ++i) {
    WF_REQUIRE(desc->send_neighbors[i] >= 0 && desc->send_neighbors[i] < comm->nranks, "wf_updater_create: send neighbour out of range");
    WF_REQUIRE(desc->send_offsets[i + 1] >= desc->send_offsets[i] && desc->send_offsets[0] == 0, "wf_updater_create: send offsets not monotone");
    u->send_nb.push_back(desc->send_neighbors[i]);
    u->send_off.push_back(desc->send_offsets[i + 1]);
  }
  for (int i = 0; i < nr; ++i) {
    WF_REQUIRE(desc->recv_neighbors[i] >= 0 && desc->recv_neighbors[i] < comm->nranks, "wf_updater_create: recv neighbour out of range");
    WF_REQUIRE(desc->recv_offsets[i + 1] >= desc->recv_offsets[i] && desc->recv_offsets[0] == 0, "wf_updater_create: recv offsets not monotone");
    u->recv_nb.push_back(desc->recv_neighbors[i]);
    u->recv_off.push_back(desc->recv_offsets[i + 1]);
  }
  u->nsend = u->send_off.back();
  u->nrecv = u->recv_off.back();
  WF_REQUIRE(u->nsend == 0 || desc->send_indices, "wf_updater_create: send_indices missing");
  WF_REQUIRE(u->nrecv == 0 || desc->ghost_positions, "wf_updater_create: ghost_positions missing");
  // every index the pack/unpack kernels dereference is checked on the host
  for (int32_t i = 0; i < u->nsend; ++i)
    WF_REQUIRE(desc->send_indices[i] >= 0 && desc->send_indices[i] < desc->ndofs, "wf_updater_create: send index out of range");
  for (int32_t i = 0; i < u->nrecv; ++i)
    WF_REQUIRE(desc->ghost_positions[i] >= 0 && desc->ghost_positions[i] < desc->ndofs, "wf_updater_create: ghost position out of range");
  if (u->nsend) {
    WF_HIP_CHECK(hipMalloc((void**)&u->d_indices, (size_t)u->nsend * sizeof(int32_t)));
    WF_HIP_CHECK(hipMemcpy(u->d_indices, desc->send_indices, (size_t)u->nsend * sizeof(int32_t), hipMemcpyHostToDevice));
    WF_HIP_CHECK(hipMalloc((void**)&u->d_send_buffer, (size_t)u->nsend * sizeof(double)));
  }
  if (u->nrecv) {
    WF_HIP_CHECK(hipMalloc((void**)&u->d_ghost_pos, (size_t)u->nrecv * sizeof(int32_t)));
    WF_HIP_CHECK(hipMemcpy(u->d_ghost_pos, desc->ghost_positions, (size_t)u->nrecv * sizeof(int32_t), hipMemcpyHostToDevice));
    WF_HIP_CHECK(hipMalloc((void**)&u->d_recv_buffer, (size_t)u->nrecv * sizeof(double)));
  }
  int prio_low = 0, prio_high = 0;
  WF_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  WF_HIP_CHECK(hipStreamCreateWithPriority(&u->comm_stream, hipStreamNonBlocking, prio_high));
  WF_HIP_CHECK(hipStreamCreateWithPriority(&u->side_stream, hipStreamNonBlocking, prio_high));
  WF_HIP_CHECK(hipStreamCreateWithPriority(&u->low_stream, hipStreamNonBlocking, prio_low));
  WF_HIP_CHECK(hipEventCreateWithFlags(&u->ev_interior, hipEventDisableTiming));
  WF_HIP_CHECK(hipEventCreateWithFlags(&u->ev_packed, hipEventDisableTiming));
  WF_HIP_CHECK(hipEventCreateWithFlags(&u->ev_done, hipEventDisableTiming));
  WF_HIP_CHECK(hipEventCreateWithFlags(&u->ev_main, hipEventDisableTiming));
  WF_HIP_CHECK(hipEventCreateWithFlags(&u->ev_side, hipEventDisableTiming));
  *out = u.release();
  return WF_OK;
}

int wf_updater_fwd_begin(wf_updater* u, const double* d_x, void* stream)
{
  WF_REQUIRE(u && d_x, "wf_updater_fwd_begin: null argument");
  return fwd_begin(u, d_x, (hipStream_t)stream, is_inline(u));
}
int wf_updater_fwd_end(wf_updater* u, double* d_x, void* stream)
{
  WF_REQUIRE(u && d_x, "wf_updater_fwd_end: null argument");
  return fwd_end(u, d_x, (hipStream_t)stream, is_inline(u));
}
int wf_updater_rev_begin(wf_updater* u, const double* d_x, void* stream)
{
  WF_REQUIRE(u && d_x, "wf_updater_rev_begin: null argument");
  return rev_begin(u, d_x, (hipStream_t)stream, is_inline(u));
}
int wf_updater_rev_end(wf_updater* u, double* d_x, void* stream)
{
  WF_REQUIRE(u && d_x, "wf_updater_rev_end: null argument");
  return rev_end(u, d_x, (hipStream_t)stream, is_inline(u));
}
int wf_updater_fwd(wf_updater* u, double* d_x, void* stream)
{
  int rc = wf_updater_fwd_begin(u, d_x, stream);
  return rc != WF_OK ? rc : wf_updater_fwd_end(u, d_x, stream);
}
int wf_updater_rev(wf_updater* u, double* d_x, void* stream)
{
  int rc = wf_updater_rev_begin(u, d_x, stream);
  return rc != WF_OK ? rc : wf_updater_rev_end(u, d_x, stream);
}

int wf_updater_info(const wf_updater* u, int* num_send, int* num_recv, int* num_send_neighbors, int* num_recv_neighbors)
{
  WF_REQUIRE(u != nullptr, "wf_updater_info: null handle");
  if (num_send) *num_send = u->nsend;
  if (num_recv) *num_recv = u->nrecv;
  if (num_send_neighbors) *num_send_neighbors = (int)u->send_nb.size();
  if (num_recv_neighbors) *num_recv_neighbors = (int)u->recv_nb.size();
  return WF_OK;
}

// internal (cg.hip): the ghost positions, for reductions over owned entries only
int wf_updater_ghosts(const wf_updater* u, const int32_t** d_ghost_pos, int32_t* nghost)
{
  WF_REQUIRE(u && d_ghost_pos && nghost, "wf_updater_ghosts: null argument");
  *d_ghost_pos = u->d_ghost_pos;
  *nghost = u->nrecv;
  return WF_OK;
}

int wf_updater_destroy(wf_updater* u)
{
  free_updater(u);
  return WF_OK;
}

// y += A x on a domain-decomposed mesh with both halo directions hidden behind the
// cells that read no ghost value:
//   side stream: update_fwd(x) -> apply(INTERFACE) -> update_rev(y)
//   `stream`   : apply(INTERIOR)
// and `stream` continues after both.  Needs wf_op_set_ghost_faces on a box operator.
int wf_op_apply_overlapped(wf_op* op, wf_updater* u, double* d_x, double* d_y, void* stream)
{
  WF_REQUIRE(op && u && d_x && d_y, "wf_op_apply_overlapped: null argument");
  wf::MarkerScope mk("wf_op_apply_overlapped");
  hipStream_t main = (hipStream_t)stream, side = u->side_stream;
  int rc;
  // Default: the halo chain (pack, RCCL, unpack, interface cells, and the same in reverse) runs on
  // the CALLER's stream and the interior cells on a low-priority stream of the updater.  The chain
  // is the longer of the two, so the caller's stream continues behind the reverse unpack with no
  // cross-stream wait on the critical path (the interior finished earlier; waiting on a signalled
  // event is free), where the mirror arrangement paid ~17 us of event latency before the next
  // kernel (rocprofv3 kernel trace of bench.py --periodic xyz).  WF_UPDATER_CHAIN_ON_SIDE selects
  // the mirror arrangement (chain on the updater's high-priority stream, interior on the caller's).
  const bool chain_on_side = (u->flags & WF_UPDATER_CHAIN_ON_SIDE) != 0;
  if (!chain_on_side) {
    WF_HIP_CHECK(hipEventRecord(u->ev_main, main));
    WF_HIP_CHECK(hipStreamWaitEvent(u->low_stream, u->ev_main, 0));
    if ((rc = wf_op_apply_part(op, d_x, d_y, WF_PART_INTERIOR, u->low_stream)) != WF_OK) return rc;
    WF_HIP_CHECK(hipEventRecord(u->ev_interior, u->low_stream));
    if ((rc = fwd_begin(u, d_x, main, true)) != WF_OK || (rc = fwd_end(u, d_x, main, true)) != WF_OK) return rc;
    if ((rc = wf_op_apply_part(op, d_x, d_y, WF_PART_INTERFACE, main)) != WF_OK) return rc;
    if ((rc = rev_begin(u, d_y, main, true)) != WF_OK || (rc = rev_end(u, d_y, main, true)) != WF_OK) return rc;
    WF_HIP_CHECK(hipStreamWaitEvent(main, u->ev_interior, 0));
    return WF_OK;
  }
  WF_HIP_CHECK(hipEventRecord(u->ev_main, main));
  WF_HIP_CHECK(hipStreamWaitEvent(side, u->ev_main, 0));
  // the side stream IS the communication stream here: exchanges are enqueued on it directly
  if ((rc = fwd_begin(u, d_x, side, true)) != WF_OK || (rc = fwd_end(u, d_x, side, true)) != WF_OK) return rc;
  if ((rc = wf_op_apply_part(op, d_x, d_y, WF_PART_INTERFACE, side)) != WF_OK) return rc;
  if ((rc = rev_begin(u, d_y, side, true)) != WF_OK || (rc = rev_end(u, d_y, side, true)) != WF_OK) return rc;
  if ((rc = wf_op_apply_part(op, d_x, d_y, WF_PART_INTERIOR, main)) != WF_OK) return rc;
  WF_HIP_CHECK(hipEventRecord(u->ev_side, side));
  WF_HIP_CHECK(hipStreamWaitEvent(main, u->ev_side, 0));
  return WF_OK;
}

}  // extern "C"
