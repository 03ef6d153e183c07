"""Domain decomposition of the box mesh and the ghost exchange.

Replaces demo/gpu_scatter_mpi/VectorUpdater.hpp:21-230 (GPU pack + CUDA-aware
MPI point-to-point over the IndexMap neighbourhood) and the
la::Vector::scatter_fwd / scatter_rev(add) calls of common/LinearGLL.hpp:164-176:
one process per GPU.  Two transports carry the same index lists:
  "native": wf_updater_* of the C ABI (csrc/comm.hip) -- a grouped ncclSend/ncclRecv
            per neighbour on an RCCL communicator created inside libwavehip, i.e.
            one message per xGMI link for the 2x2x2 partition; the default for
            device vectors;
  "torch" : ONE torch.distributed all_to_all_single per update (gloo on CPU tensors
            for the multi-process CPU tests, or nccl); pack/unpack are the libwavehip
            gather/scatter kernels either way.

Partition: Cartesian px x py x pz (the idea of decompose3d /
compute_cartesian_indices in demo/gpu_cg/mesh.hpp:37-63), rank = rz + pz*(ry + py*rx).
Ownership: a lattice point shared by several ranks belongs to the lowest one, so
a rank's ghosts are the lower planes (I = 0, J = 0, K = 0) of its local lattice
where a lower neighbour exists.  Local vectors use the local lattice numbering
(ghosts interleaved, not appended) because the structured kernels address the
lattice implicitly; the owned/ghost split is carried by the index lists."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .box import BoxMesh, FunctionSpace, IndexMap, create_box, create_functionspace


def decompose3d(nproc: int):
    """Split nproc into px >= py >= pz, as balanced as possible
    (1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2; cf. demo/gpu_cg/mesh.hpp:37-48)."""
    best = None
    for a in range(1, nproc + 1):
        if nproc % a:
            continue
        for b in range(1, nproc // a + 1):
            if (nproc // a) % b:
                continue
            c = nproc // a // b
            dims = tuple(sorted((a, b, c), reverse=True))
            score = (dims[0] - dims[2], dims[0])
            if best is None or score < best[0]:
                best = (score, dims)
    return best[1]


def rank_coords(rank: int, procs):
    """demo/gpu_cg/mesh.hpp:52-63 compute_cartesian_indices (z fastest)."""
    px, py, pz = procs
    return rank // (py * pz), (rank // pz) % py, rank % pz


def coords_rank(c, procs):
    px, py, pz = procs
    return c[2] + pz * (c[1] + py * c[0])


@dataclass
class BoxPartition:
    procs: tuple
    rank: int
    coords: tuple
    n_local: tuple              # cells per direction on this rank
    degree: int
    mesh: BoxMesh
    V: FunctionSpace
    size_global: int            # global number of (owned) dofs
    owned_lo: tuple             # first owned lattice index per axis (0 or 1)
    # neighbour lists, keyed by neighbour rank (ascending); int32 local lattice indices
    send_fwd: dict = field(default_factory=dict)   # owned dofs the neighbour holds as ghosts
    recv_fwd: dict = field(default_factory=dict)   # my ghosts owned by the neighbour
    periodic: tuple = (False, False, False)        # axes whose upper face is identified with the lower one

    @property
    def num_owned(self) -> int:
        NX, NY, NZ = self.V.lattice
        return (NX - self.owned_lo[0]) * (NY - self.owned_lo[1]) * (NZ - self.owned_lo[2])

    def owned_mask(self) -> np.ndarray:
        NX, NY, NZ = self.V.lattice
        m = np.ones((NZ, NY, NX), dtype=bool)
        if self.owned_lo[0]:
            m[:, :, 0] = False
        if self.owned_lo[1]:
            m[:, 0, :] = False
        if self.owned_lo[2]:
            m[0, :, :] = False
        return m.reshape(-1)


def _axis_range(kind: int, lo: int, hi: int):
    """kind -1: the lower plane (index 0); +1: the upper plane (index hi); 0: the owned range."""
    if kind < 0:
        return np.array([0])
    if kind > 0:
        return np.array([hi])
    return np.arange(lo, hi + 1)


def create_distributed_box(n, degree: int, nproc: int, rank: int, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0),
                           perturb: float = 0.0, seed: int = 42, build_dofmap: bool = False,
                           periodic=(False, False, False)) -> BoxPartition:
    """Weak-scaled box: n (int or 3-tuple) cells per direction PER RANK; the
    global mesh has (px*nx, py*ny, pz*nz) cells on [lo, hi].
    periodic[a] identifies the upper face of axis a with the lower one: the lower
    plane of EVERY rank is then a ghost plane and the neighbour relation wraps
    around, so a rank can be its own neighbour (with one rank per axis the exchange
    is a send/recv to self -- used to run the RCCL path on a single GPU)."""
    if np.isscalar(n):
        n = (int(n),) * 3
    procs = decompose3d(nproc)
    c = rank_coords(rank, procs)
    p = degree
    gn = tuple(procs[a] * n[a] for a in range(3))
    # local mesh = the rank's slab of the global vertex lattice (global coordinates;
    # the perturbation is drawn on the global mesh so that ranks agree on shared vertices)
    gmesh_x = None
    if perturb > 0.0:
        gmesh_x = create_box(gn, lo, hi, perturb, seed).x.reshape(gn[2] + 1, gn[1] + 1, gn[0] + 1, 3)
    h = [(hi[a] - lo[a]) / gn[a] for a in range(3)]
    llo = tuple(lo[a] + h[a] * n[a] * c[a] for a in range(3))
    lhi = tuple(lo[a] + h[a] * n[a] * (c[a] + 1) for a in range(3))
    mesh = create_box(n, llo, lhi)
    if gmesh_x is not None:
        sl = tuple(slice(c[a] * n[a], (c[a] + 1) * n[a] + 1) for a in range(3))
        mesh.x = np.ascontiguousarray(gmesh_x[sl[2], sl[1], sl[0]].reshape(-1, 3))
    V = create_functionspace(mesh, p, build_dofmap=build_dofmap)
    NX, NY, NZ = V.lattice
    periodic = tuple(bool(v) for v in periodic)
    owned_lo = tuple(1 if (c[a] > 0 or periodic[a]) else 0 for a in range(3))
    part = BoxPartition(procs, rank, c, tuple(n), p, mesh, V,
                        size_global=int(np.prod([p * gn[a] + (0 if periodic[a] else 1) for a in range(3)])),
                        owned_lo=owned_lo, periodic=periodic)
    V.index_map = IndexMap(NX * NY * NZ, 0, part.size_global)   # whole local lattice; see module docstring
    hi_idx = (NX - 1, NY - 1, NZ - 1)

    def lattice_indices(kinds):
        ix = _axis_range(kinds[0], owned_lo[0], hi_idx[0])
        iy = _axis_range(kinds[1], owned_lo[1], hi_idx[1])
        iz = _axis_range(kinds[2], owned_lo[2], hi_idx[2])
        K, J, I = np.meshgrid(iz, iy, ix, indexing="ij")
        return (I + NX * (J + NY * K)).reshape(-1).astype(np.int32)

    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                if (dx, dy, dz) == (0, 0, 0):
                    continue
                d = (dx, dy, dz)
                wrap = lambda q: tuple(q[a] % procs[a] if periodic[a] else q[a] for a in range(3))
                # upper neighbour +d holds my upper plane(s) as ghosts
                up = wrap(tuple(c[a] + d[a] for a in range(3)))
                if all(0 <= up[a] < procs[a] for a in range(3)):
                    _append(part.send_fwd, coords_rank(up, procs), lattice_indices(d))
                # lower neighbour -d owns my lower plane(s)
                dn = wrap(tuple(c[a] - d[a] for a in range(3)))
                if all(0 <= dn[a] < procs[a] for a in range(3)):
                    _append(part.recv_fwd, coords_rank(dn, procs), lattice_indices(tuple(-v for v in d)))
    return part


def _append(lists: dict, nb: int, idx: np.ndarray):
    """Several directions can lead to the same neighbour on a periodic partition
    (always to the rank itself when an axis has one rank): their segments are
    concatenated in direction order, which is the same on the sending and the
    receiving side."""
    lists[nb] = np.concatenate([lists[nb], idx]) if nb in lists else idx


def boundary_tags(part: BoxPartition) -> dict:
    """Facet tags of the rank's local box faces under the cfg1 convention
    (SURVEY 8d): global face x = lo -> tag 1 (Gamma_1), every other global face
    -> tag 2 (Gamma_2); interfaces between ranks carry no tag.
    Key = local face 2*axis + side."""
    tags = {}
    for axis in range(3):
        if part.periodic[axis]:
            continue
        if part.coords[axis] == 0:
            tags[2 * axis] = 1 if axis == 0 else 2
        if part.coords[axis] == part.procs[axis] - 1:
            tags[2 * axis + 1] = 2
    return tags


class HipKernels:
    """Pack/unpack through libwavehip (no fallback)."""

    @staticmethod
    def gather(idx, src, out):
        from .operators import gather
        gather(idx.numel(), idx, src, out)

    @staticmethod
    def scatter_set(idx, src, out):
        from .operators import scatter_set
        scatter_set(idx.numel(), idx, src, out)

    @staticmethod
    def scatter_add(idx, src, out):
        from .operators import scatter
        scatter(idx.numel(), idx, src, out)


class VectorUpdater:
    """VectorUpdater<T, Alloc> of demo/gpu_scatter_mpi/VectorUpdater.hpp:21-230.

    update_fwd(x): owners -> ghosts (VectorUpdater.hpp:148-152);
    update_rev(x): ghosts -> owners, accumulating (VectorUpdater.hpp:204-208).
    scatter_fwd / scatter_rev are the la::Vector spellings used by
    common/LinearGLL.hpp.  The _begin/_end split of the reference is kept:
    begin packs and posts the exchange, end waits and unpacks.

    transport "native" (default for device vectors unless the process group is gloo):
    wf_updater_* over an RCCL communicator (`comm`, created from torch.distributed's
    ranks when not given).  transport "torch": all_to_all_single on `group`."""

    def __init__(self, part: BoxPartition, device=None, group=None, kernels=None, transport: str | None = None,
                 comm=None):
        import torch
        import torch.distributed as dist
        self.part = part
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device("cpu") if device is None else device
        self.kernels = HipKernels if kernels is None else kernels
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        if transport is None:
            transport = "native" if (self.device.type == "cuda" and self.backend != "gloo" and kernels is None) else "torch"
        if transport not in ("native", "torch"):
            raise ValueError("transport must be 'native' or 'torch'")
        self.transport = transport
        # neighbours in ascending rank order; displacements and sizes per neighbour (VectorUpdater.hpp:34-46)
        self.send_neighbors = sorted(part.send_fwd)
        self.recv_neighbors = sorted(part.recv_fwd)
        cat = lambda L: np.concatenate(L).astype(np.int32) if L else np.zeros(0, dtype=np.int32)
        self.h_indices = cat([part.send_fwd[r] for r in self.send_neighbors])       # scatter_fwd_indices
        self.h_ghost_pos = cat([part.recv_fwd[r] for r in self.recv_neighbors])     # ghost positions
        self.active = bool(self.send_neighbors or self.recv_neighbors)
        self._work = None
        if transport == "native":
            self._init_native(comm)
            return
        if self.world == 1 and self.active:
            raise RuntimeError("a self-neighbour (periodic) exchange on one rank needs transport='native'")
        self.send_sizes = [int(part.send_fwd[r].size) if r in part.send_fwd else 0 for r in range(self.world)]
        self.recv_sizes = [int(part.recv_fwd[r].size) if r in part.recv_fwd else 0 for r in range(self.world)]
        self.d_indices = torch.from_numpy(self.h_indices).to(self.device)
        self.d_ghost_pos = torch.from_numpy(self.h_ghost_pos).to(self.device)
        # zero-length exchanges (a rank with no ghosts / no upper neighbour) keep a valid allocation behind the view
        self.d_send_buffer = torch.zeros(max(self.d_indices.numel(), 1), dtype=torch.float64, device=self.device)[: self.d_indices.numel()]
        self.d_recv_buffer = torch.zeros(max(self.d_ghost_pos.numel(), 1), dtype=torch.float64, device=self.device)[: self.d_ghost_pos.numel()]
        # a transport that cannot read device memory (gloo) is staged through the host
        self.staged = self.backend == "gloo" and self.device.type == "cuda"
        if self.staged:
            self.h_send = torch.zeros(max(self.d_indices.numel(), self.d_ghost_pos.numel()), dtype=torch.float64).pin_memory()
            self.h_recv = torch.zeros_like(self.h_send).pin_memory()

    # -- native transport (csrc/comm.hip) ----------------------------------------
    def _init_native(self, comm):
        import ctypes
        from ctypes import POINTER, c_int, c_int32, c_void_p

        from . import _lib
        from .comm import Comm
        if self.device.type != "cuda":
            raise RuntimeError("transport='native' exchanges device vectors (RCCL)")
        self.comm = comm
        if self.comm is None and self.active:
            self.comm = Comm.from_torch_distributed(self.group)
        d = _lib.UpdaterDesc()
        d.ndofs = int(self.part.V.ndofs)
        snb = np.asarray(self.send_neighbors, dtype=np.intc)
        rnb = np.asarray(self.recv_neighbors, dtype=np.intc)
        soff = np.concatenate([[0], np.cumsum([self.part.send_fwd[r].size for r in self.send_neighbors])]).astype(np.int32)
        roff = np.concatenate([[0], np.cumsum([self.part.recv_fwd[r].size for r in self.recv_neighbors])]).astype(np.int32)
        ip = lambda a: a.ctypes.data_as(POINTER(c_int32))
        d.num_send_neighbors, d.num_recv_neighbors = len(snb), len(rnb)
        d.send_neighbors = snb.ctypes.data_as(POINTER(c_int))
        d.recv_neighbors = rnb.ctypes.data_as(POINTER(c_int))
        d.send_offsets, d.recv_offsets = ip(soff), ip(roff)
        d.send_indices, d.ghost_positions = ip(self.h_indices), ip(self.h_ghost_pos)
        d.flags = _lib.WF_UPDATER_DEFAULT
        self._h = c_void_p()
        _lib.check(_lib.lib().wf_updater_create(self.comm._h if self.comm is not None else None, ctypes.byref(d),
                                                ctypes.byref(self._h)))

    def _native(self, fn: str, x):
        from . import _lib
        from .operators import _ptr, _stream
        _lib.check(getattr(_lib.lib(), fn)(self._h, _ptr(x), _stream(x)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            from . import _lib
            _lib.lib().wf_updater_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- torch transport ------------------------------------------------------------
    def _exchange(self, out, out_sizes, inp, in_sizes):
        import torch.distributed as dist
        if self.staged:
            import torch
            hs, hr = self.h_send[: inp.numel()], self.h_recv[: out.numel()]
            hs.copy_(inp)
            torch.cuda.current_stream().synchronize()
            dist.all_to_all_single(hr, hs, out_sizes, in_sizes, group=self.group)
            out.copy_(hr, non_blocking=True)
            return None
        return dist.all_to_all_single(out, inp, out_sizes, in_sizes, group=self.group, async_op=True)

    def _wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None

    # -- forward: owners -> ghosts --------------------------------------------
    def update_fwd_begin(self, x):
        if not self.active:
            return
        if self.transport == "native":
            return self._native("wf_updater_fwd_begin", x)
        self.kernels.gather(self.d_indices, x, self.d_send_buffer)                  # VectorUpdater.hpp:110-111
        self._work = self._exchange(self.d_recv_buffer, self.recv_sizes, self.d_send_buffer, self.send_sizes)

    def update_fwd_end(self, x):
        if not self.active:
            return
        if self.transport == "native":
            return self._native("wf_updater_fwd_end", x)
        self._wait()
        self.kernels.scatter_set(self.d_ghost_pos, self.d_recv_buffer, x)           # VectorUpdater.hpp:139-142

    def update_fwd(self, x):
        self.update_fwd_begin(x)
        self.update_fwd_end(x)

    # -- reverse: ghosts -> owners (add) ---------------------------------------
    def update_rev_begin(self, x):
        if not self.active:
            return
        if self.transport == "native":
            return self._native("wf_updater_rev_begin", x)
        self.kernels.gather(self.d_ghost_pos, x, self.d_recv_buffer)                # VectorUpdater.hpp:165-168
        self._work = self._exchange(self.d_send_buffer, self.send_sizes, self.d_recv_buffer, self.recv_sizes)

    def update_rev_end(self, x):
        if not self.active:
            return
        if self.transport == "native":
            return self._native("wf_updater_rev_end", x)
        self._wait()
        self.kernels.scatter_add(self.d_indices, self.d_send_buffer, x)             # VectorUpdater.hpp:196-198

    def update_rev(self, x):
        self.update_rev_begin(x)
        self.update_rev_end(x)

    scatter_fwd = update_fwd
    scatter_rev = update_rev


_SIDE_STREAMS = {}


def overlapped_apply(op, updater: VectorUpdater, x, y):
    """y += A x on a domain-decomposed mesh with BOTH halo directions hidden
    (LinearGLL.hpp:164-176 = scatter_fwd(x); apply; scatter_rev(y)):

        side stream : update_fwd(x) -> apply(INTERFACE) -> update_rev(y)
        main stream : apply(INTERIOR)                       (reads no ghost value)

    The interior part is one launch (splitting it further costs whole rounds of
    workgroups: 3 launches took 0.36 ms against 0.26 ms for the unsplit operator at
    cfg2), the interface cells and the two exchanges run beside it on a second HIP
    stream and fill its tail.  Requires op.set_ghost_faces(...) == True.
    Native transport: the whole sequence is wf_op_apply_overlapped of the C ABI."""
    import torch
    from ._lib import WF_PART_INTERFACE, WF_PART_INTERIOR
    if not x.is_cuda:
        raise RuntimeError("overlapped_apply needs device vectors")
    if updater.transport == "native":
        from . import _lib
        from .operators import _ptr, _stream
        _lib.check(_lib.lib().wf_op_apply_overlapped(op._h, updater._h, _ptr(x), _ptr(y), _stream(x)))
        return
    main = torch.cuda.current_stream(x.device)
    side = _SIDE_STREAMS.get(x.device)
    if side is None:
        side = _SIDE_STREAMS[x.device] = torch.cuda.Stream(device=x.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        updater.update_fwd(x)
        op.apply_part(x, y, WF_PART_INTERFACE)
        updater.update_rev(y)
    op.apply_part(x, y, WF_PART_INTERIOR)
    main.wait_stream(side)


def owned_boundary_set(updater: VectorUpdater, V, idx, m, device):
    """A boundary dof set of a domain-decomposed mesh (indices, rank-local collocated facet masses)
    reduced to OWNED dofs with fully assembled masses: the rank-local masses are accumulated to
    their owners once at setup (scatter_rev), so that the boundary term b[i] += s m[i] v[i] is
    applied by the owner alone and needs no ghost value of v -- the second forward update per
    stage of the reference (LinearGLL.hpp:167) disappears."""
    import torch
    idx = np.asarray(idx)
    dense = torch.zeros(V.ndofs, dtype=torch.float64, device=device)
    if idx.size:
        dense[torch.from_numpy(idx.astype(np.int64)).to(device)] = torch.from_numpy(np.ascontiguousarray(m, dtype=np.float64)).to(device)
    updater.scatter_rev(dense)
    h = dense.cpu().numpy()
    sel = np.nonzero((h != 0.0) & updater.part.owned_mask())[0].astype(np.int32)
    return sel, h[sel]


def owned_boundary(updater: VectorUpdater, V, tags, tag: int, device):
    """owned_boundary_set for the faces of a box partition carrying `tag`."""
    from .linear_gll import facet_lumped_mass
    idx, m = facet_lumped_mass(V, tags, tag)
    return owned_boundary_set(updater, V, idx, m, device)
