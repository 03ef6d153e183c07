"""N > 1 path on CPU: world_size 2/4/8 gloo processes exercise the Cartesian
partition, the owner/ghost index lists and the all_to_all neighbour exchange of
wave_fenics_amd.distributed; the local operator is the CPU oracle on each rank's
sub-mesh, the answer is the oracle on the global mesh."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_decompose():
    from wave_fenics_amd.distributed import decompose3d, rank_coords, coords_rank
    assert decompose3d(1) == (1, 1, 1) and decompose3d(2) == (2, 1, 1)
    assert decompose3d(4) == (2, 2, 1) and decompose3d(8) == (2, 2, 2)
    assert decompose3d(6) == (3, 2, 1) and decompose3d(12) == (3, 2, 2)
    for procs in [(2, 2, 2), (3, 2, 1)]:
        for r in range(int(np.prod(procs))):
            assert coords_rank(rank_coords(r, procs), procs) == r


def test_partition_lists_cover_ghosts():
    """Every ghost is received exactly once; send/recv lists of neighbours match
    in length and in the global points they name."""
    from dist_helpers import local_to_global
    from wave_fenics_amd.distributed import create_distributed_box
    for world, n, p in [(8, (2, 2, 2), 2), (4, (2, 1, 2), 3), (2, (1, 2, 2), 4), (12, (1, 1, 1), 2)]:
        parts = [create_distributed_box(n, p, world, r) for r in range(world)]
        total_owned = 0
        for r, part in enumerate(parts):
            l2g = local_to_global(part)
            ghost = ~part.owned_mask()
            recv_all = np.concatenate([v for v in part.recv_fwd.values()]) if part.recv_fwd else np.zeros(0, int)
            assert np.array_equal(np.sort(recv_all), np.nonzero(ghost)[0])
            total_owned += part.num_owned
            for nb, idx in part.recv_fwd.items():
                other = parts[nb]
                assert r in other.send_fwd and other.send_fwd[r].size == idx.size
                assert np.array_equal(local_to_global(other)[other.send_fwd[r]], l2g[idx])
                assert other.owned_mask()[other.send_fwd[r]].all()
        assert total_owned == parts[0].size_global
        if world == 8:
            # at most one message per xGMI link (7 peers); rank 0 talks to all 7
            assert all(len(pt.send_fwd) + len(pt.recv_fwd) <= 7 for pt in parts) and len(parts[0].send_fwd) == 7


def test_periodic_partition_lists():
    """Periodic partitions (a rank may be its own neighbour, several directions may
    lead to the same neighbour): every ghost is received exactly once, the k-th entry
    a rank sends to a neighbour is the k-th entry that neighbour receives from it."""
    from dist_helpers import local_to_global
    from wave_fenics_amd.distributed import create_distributed_box
    cases = [(1, (2, 2, 2), 2, (True, False, False)), (1, (2, 1, 2), 3, (True, True, True)),
             (2, (2, 2, 1), 2, (True, True, False)), (4, (1, 2, 2), 2, (True, True, True)),
             (8, (1, 1, 1), 2, (True, False, True)), (2, (2, 2, 2), 1, (False, True, False))]
    for world, n, p, per in cases:
        parts = [create_distributed_box(n, p, world, r, periodic=per) for r in range(world)]
        assert sum(pt.num_owned for pt in parts) == parts[0].size_global
        owned_global = []
        for r, part in enumerate(parts):
            l2g = local_to_global(part)
            ghost = ~part.owned_mask()
            recv_all = np.concatenate(list(part.recv_fwd.values())) if part.recv_fwd else np.zeros(0, int)
            assert np.array_equal(np.sort(recv_all), np.nonzero(ghost)[0])
            owned_global.append(l2g[part.owned_mask()])
            for nb, idx in part.recv_fwd.items():
                other = parts[nb]
                assert r in other.send_fwd and other.send_fwd[r].size == idx.size
                assert np.array_equal(local_to_global(other)[other.send_fwd[r]], l2g[idx])
                assert other.owned_mask()[other.send_fwd[r]].all()
        allg = np.concatenate(owned_global)
        assert np.array_equal(np.sort(allg), np.arange(parts[0].size_global))      # every global dof owned exactly once


def test_cxx_partition_matches_python(tmp_path):
    """include/wavehip_box.hpp create_distributed_box (the C++ host side of
    VectorUpdater) names the same neighbours and the same dofs in the same order as
    wave_fenics_amd.distributed, for plain and periodic partitions."""
    import subprocess
    from wave_fenics_amd import build
    from wave_fenics_amd.distributed import boundary_tags, create_distributed_box
    build.build()
    exe = str(tmp_path / "partition_dump")
    libdir = os.path.join(ROOT, "wave_fenics_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "partition_dump.cpp"), "-o", exe,
                           "-L", libdir, "-lwavehip", f"-Wl,-rpath,{libdir}"])
    cases = [(1, (2, 2, 2), 2, (1, 0, 0)), (1, (2, 1, 2), 3, (1, 1, 1)), (2, (2, 2, 1), 2, (1, 1, 0)),
             (4, (1, 2, 2), 2, (0, 0, 0)), (8, (1, 2, 1), 2, (0, 0, 0)), (8, (1, 1, 1), 3, (1, 0, 1)),
             (6, (2, 1, 1), 1, (0, 1, 0))]
    for world, n, p, per in cases:
        for rank in range(world):
            part = create_distributed_box(n, p, world, rank, periodic=per)
            out = subprocess.run([exe] + [str(v) for v in (*n, p, world, rank, *per)], capture_output=True, text=True,
                                 timeout=60)
            assert out.returncode == 0, out.stderr
            lines = out.stdout.strip().splitlines()
            head = lines[0].split()
            assert tuple(int(v) for v in head[1:4]) == tuple(part.procs)
            assert tuple(int(v) for v in head[5:8]) == tuple(part.coords)
            assert tuple(int(v) for v in head[9:12]) == tuple(part.owned_lo)
            assert int(head[13]) == part.size_global and int(head[15]) == part.num_owned
            send = {int(l.split()[1]): np.array(l.split()[2:], dtype=np.int32) for l in lines if l.startswith("send")}
            recv = {int(l.split()[1]): np.array(l.split()[2:], dtype=np.int32) for l in lines if l.startswith("recv")}
            assert sorted(send) == sorted(part.send_fwd) and sorted(recv) == sorted(part.recv_fwd)
            for nb in send:
                assert np.array_equal(send[nb], part.send_fwd[nb])
            for nb in recv:
                assert np.array_equal(recv[nb], part.recv_fwd[nb])
            tags = {int(t.split(":")[0]): int(t.split(":")[1]) for t in lines[-1].split()[1:]}
            assert tags == boundary_tags(part)


def _worker(rank, world, port, n, p, perturb, q):
    try:
        import torch
        import torch.distributed as dist
        from dist_helpers import TorchIndexKernels, init_pg, local_to_global
        from oracle import wave_oracle as o
        from wave_fenics_amd.distributed import VectorUpdater, create_distributed_box
        init_pg(rank, world, port)
        part = create_distributed_box(n, p, world, rank, perturb=perturb)
        vu = VectorUpdater(part, kernels=TorchIndexKernels)
        l2g = local_to_global(part)
        owned = part.owned_mask()
        gn = tuple(part.procs[a] * n[a] for a in range(3))
        gm = o.create_box(gn, p, perturb=perturb)
        assert np.allclose(gm.x.reshape(gn[2] + 1, gn[1] + 1, gn[0] + 1, 3)[
            part.coords[2] * n[2]:(part.coords[2] + 1) * n[2] + 1,
            part.coords[1] * n[1]:(part.coords[1] + 1) * n[1] + 1,
            part.coords[0] * n[0]:(part.coords[0] + 1) * n[0] + 1].reshape(-1, 3), part.mesh.x, atol=1e-15)
        rng = np.random.default_rng(77)
        xg = rng.uniform(-1, 1, gm.ndofs)
        # (i) the reference's own procedure: fill with the rank id, ghosts must show the owner's id
        #     (demo/gpu_scatter_mpi/main.cpp:97)
        xr = torch.full((l2g.size,), float(rank), dtype=torch.float64)
        vu.update_fwd(xr)
        for nb, idx in part.recv_fwd.items():
            assert np.all(xr.numpy()[idx] == float(nb))
        # (ii) forward update reproduces the global vector on owned + ghost entries, bit for bit
        xl = np.zeros(l2g.size)
        xl[owned] = xg[l2g[owned]]
        xt = torch.from_numpy(xl)
        vu.update_fwd(xt)
        assert np.array_equal(xt.numpy(), xg[l2g])
        # (iii) local operator + reverse (add) update == global operator on owned entries
        lm = o.create_box(n, p)
        lm.x = part.mesh.x.copy()
        Kl = o.StiffnessOperator(lm, p)
        yl = torch.zeros(l2g.size, dtype=torch.float64)
        Kl(xt.numpy(), yl.numpy())
        vu.update_rev(yl)
        Kg = o.StiffnessOperator(gm, p)
        yg = np.zeros(gm.ndofs)
        Kg(xg, yg)
        err = np.abs(yl.numpy()[owned] - yg[l2g[owned]]).max() / np.abs(yg).max()
        # (iv) lumped mass: m = M 1, scatter_rev(add)  (LinearGLL.hpp:105-110)
        Ml = o.MassOperatorCPU(lm, p)
        ml = torch.zeros(l2g.size, dtype=torch.float64)
        Ml(np.ones(l2g.size), ml.numpy())
        vu.scatter_rev(ml)
        mg = np.zeros(gm.ndofs)
        o.MassOperatorCPU(gm, p)(np.ones(gm.ndofs), mg)
        errm = np.abs(ml.numpy()[owned] - mg[l2g[owned]]).max() / mg.max()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, float(err), float(errm), None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, None, None, traceback.format_exc()))


@pytest.mark.parametrize("world,n,p,perturb", [(2, (2, 2, 2), 2, 0.2), (4, (2, 2, 1), 3, 0.2), (8, (1, 2, 1), 2, 0.15),
                                               (2, (1, 1, 2), 4, 0.0)])
def test_ghost_exchange_gloo(world, n, p, perturb):
    import torch.multiprocessing as mp
    from dist_helpers import free_port
    from oracle import wave_oracle
    wave_oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, p, perturb, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
    for rank, err, errm, tb in res:
        assert tb is None, f"rank {rank} failed:\n{tb}"
        assert err <= 1e-12 and errm <= 1e-13, (rank, err, errm)
