"""CPU (gloo, world_size 2 and 3) tests of the z-slab decomposition.

No GPU here, so the HIP kernels cannot run; what these tests pin is the decomposition itself --
which ghost planes are refreshed when, and how wide -- by replaying the library's orchestration
(fluidsolver_amd/csrc/fs_exchange.hip: ensure_advect_ghosts / ensure_velocity_ghosts /
run_jacobi's split passes with one exchange of 4 planes after each or, on slabs of 25 planes and more,
its pairs of passes with one exchange of 8 planes per pair, extended divergence range) with the CPU
oracle's kernels.  Every rank keeps full-size arrays but only its owned + ghost planes hold data;
everything else is poisoned with NaN after each exchange, so any read beyond the ghost planes,
and any stale ghost, shows up as a mismatch against the single-domain oracle.  The same file
tests the host-staged transport callback of fluidsolver_amd.slab between real processes.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GHOST = 4      # fs_solver.hip (create_common): ghost planes of a slab
BLOCK = 4      # fsk::jacobi_block(): sweeps fused per pass


def test_slab_range_partitions_the_grid():
    from fluidsolver_amd.slab import slab_range
    for nz in (8, 31, 64, 513):
        for world in (1, 2, 3, 8):
            if nz // world < 1:
                continue
            z = 0
            sizes = []
            for r in range(world):
                z0, n = slab_range(nz, r, world)
                assert z0 == z and n >= 1
                z += n
                sizes.append(n)
            assert z == nz and max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------ slab replay with gloo

def _exchange(dist, arrays, planes, z0, z1, rank, world, face_flags):
    """Ghost refresh of `planes` planes for every array, same ranges as make_seg()."""
    import torch
    reqs, recvs = [], []
    for a, face in zip(arrays, face_flags):
        f = 1 if face else 0
        if rank > 0:
            send = torch.from_numpy(np.ascontiguousarray(a[z0 + f:z0 + f + planes]))
            recv = torch.empty_like(send)
            reqs += [dist.P2POp(dist.isend, send, rank - 1), dist.P2POp(dist.irecv, recv, rank - 1)]
            recvs.append((a, slice(z0 - planes, z0), recv))
        if rank < world - 1:
            send = torch.from_numpy(np.ascontiguousarray(a[z1 - planes:z1]))
            recv = torch.empty_like(send)
            reqs += [dist.P2POp(dist.isend, send, rank + 1), dist.P2POp(dist.irecv, recv, rank + 1)]
            recvs.append((a, slice(z1 + f, z1 + f + planes), recv))
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
    for a, sl, recv in recvs:
        a[sl] = recv.numpy()


def _poison(a, z0, z1, face, rank, world, depth=GHOST):
    lo = 0 if rank == 0 else z0 - depth
    hi = a.shape[0] if rank == world - 1 else z1 + (1 if face else 0) + depth
    a[:max(lo, 0)] = np.nan
    a[hi:] = np.nan


def _slab_worker(rank, world, port, dims, steps, k, result):
    # ghost planes held for the pressure, its divergence and the velocity behind it (create_common)
    pghost = 2 * BLOCK if dims[2] // world >= 2 * BLOCK + 1 else GHOST
    pairs = pghost >= 2 * BLOCK and dims[2] // world >= 6 * BLOCK + 1
    import torch.distributed as dist
    import oracle
    from fluidsolver_amd.slab import slab_range
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        nx, ny, nz = dims
        kw = dict(jacobi_iters=k, source_y=min(10.0, ny / 3.0), source_radius=min(7.0, min(dims) / 4.0))
        par = oracle.default_params(nx, ny, nz, **kw)
        ref = oracle.Oracle(par)
        z0, nzl = slab_range(nz, rank, world)
        z1 = z0 + nzl
        F = np.float32
        ux = np.zeros((nz, ny, nx + 1), F); uy = np.zeros((nz, ny + 1, nx), F); uz = np.zeros((nz + 1, ny, nx), F)
        rho = np.zeros((nz, ny, nx), F); T = np.zeros((nz, ny, nx), F); p = np.zeros((nz, ny, nx), F)

        def own(a, new, face=False):
            """owned planes of a freshly computed array; every other plane keeps its stale value
            (finite in the ghost planes, NaN beyond them), as the device buffers do"""
            out = a.copy()
            hi = z1 + (1 if face else 0)
            out[z0:hi] = new[z0:hi]
            return out

        def refresh(arrs, planes, faces):
            _exchange(dist, arrs, planes, z0, z1, rank, world, faces)
            for a, f in zip(arrs, faces):
                _poison(a, z0, z1, f, rank, world, planes)

        ok = True
        p_ghost = pghost     # all-zero initial pressure: the ghost planes are valid
        for step in range(steps):
            ref.step()
            oracle.source(rho, T, par.source_x, par.source_y, par.source_z, par.source_radius, par.rho0, par.temperature0)
            oracle.force_uy(uy, T, par.dt, par.temperature_env)
            refresh([ux, uy, uz, T, rho], GHOST, [False, False, True, False, False])   # ensure_advect_ghosts
            nux = own(ux, oracle.semilag(ux, ux, uy, uz, par.dt, 1))
            nuy = own(uy, oracle.semilag(uy, ux, uy, uz, par.dt, 2))
            nuz = own(uz, oracle.semilag(uz, ux, uy, uz, par.dt, 3), face=True)
            ux, uy, uz = nux, nuy, nuz
            refresh([ux, uy, uz], pghost, [False, False, True])                          # new velocity: pghost planes
            T = own(T, oracle.semilag(T, ux, uy, uz, par.dt, 0))
            rho = own(rho, oracle.semilag(rho, ux, uy, uz, par.dt, 0))
            div = oracle.divergence(ux, uy, uz)
            lo = 0 if rank == 0 else z0 - (pghost - 1)
            hi = nz if rank == world - 1 else z1 + (pghost - 1)
            div[:lo] = np.nan; div[hi:] = np.nan                                         # extended divergence range
            left = k
            while pairs and left >= 2 * BLOCK:
                if p_ghost < 2 * BLOCK:
                    refresh([p], 2 * BLOCK, [False])                                     # ensure_p_ghosts(2T)
                # pass A: the slab and BLOCK planes beyond it; pass B: the slab
                for reach in (2 * BLOCK, BLOCK):
                    plo = 0 if rank == 0 else z0 - reach
                    phi = nz if rank == world - 1 else z1 + reach
                    q = p.copy(); q[:plo] = np.nan; q[phi:] = np.nan
                    for _ in range(BLOCK):
                        q, _r = oracle.jacobi(q, div, want_r=False)
                    keep = reach - BLOCK
                    olo = 0 if rank == 0 else z0 - keep
                    ohi = nz if rank == world - 1 else z1 + keep
                    p = p.copy(); p[olo:ohi] = q[olo:ohi]
                    p[:olo] = np.nan; p[ohi:] = np.nan
                refresh([p], 2 * BLOCK, [False])                                         # one exchange per pair
                p_ghost = 2 * BLOCK
                left -= 2 * BLOCK
            while left > 0:
                t = min(BLOCK, left)
                if p_ghost < t:
                    refresh([p], t, [False])                                             # ensure_p_ghosts(t)
                plo = 0 if rank == 0 else z0 - t
                phi = nz if rank == world - 1 else z1 + t
                q = p.copy(); q[:plo] = np.nan; q[phi:] = np.nan
                for _ in range(t):
                    q, _r = oracle.jacobi(q, div, want_r=False)
                p = own(p, q)
                # the boundary layers of the new iterate travel while the interior is computed
                refresh([p], BLOCK, [False])
                p_ghost = BLOCK
                left -= t
            assert p_ghost >= 1                                                          # ensure_p_ghosts(1): nothing to do
            gx, gy, gz = ux.copy(), uy.copy(), uz.copy()
            oracle.gradient(gx, gy, gz, p)
            ux, uy, uz = own(ux, gx), own(uy, gy), own(uz, gz, face=True)
            for name, mine, fid, face in (("ux", ux, 0, 0), ("uy", uy, 1, 0), ("uz", uz, 2, 1), ("rho", rho, 3, 0),
                                          ("T", T, 4, 0), ("p", p, 5, 0)):
                want = ref.get(fid)[z0:z1 + face]
                got = mine[z0:z1 + face]
                if not np.array_equal(got, want):
                    print("rank %d step %d %s: %d mismatches, %d NaN" % (rank, step + 1, name, int((got != want).sum()),
                                                                        int(np.isnan(got).sum())))
                    ok = False
        result[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


def _host_callback_worker(rank, world, port, result):
    """The Python exchange callback the host-staged transport hands to fs_create_slab, called the
    way the library calls it (raw pointers), between real processes."""
    import torch.distributed as dist
    from fluidsolver_amd.slab import FS_XCHG_HALO, FS_XCHG_SUM_F64, SlabSolver
    import fluidsolver_amd as fs
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        s = SlabSolver(fs.default_params(8, 8, 8 * world), rank, world, transport="host")
        cb = s._host_callback()
        n = 4096
        bufs = [np.full(n, 10 * rank + q, np.uint8) for q in range(4)]   # send_lo, recv_lo, send_hi, recv_hi
        ptr = [b.ctypes.data_as(C.c_void_p) for b in bufs]
        lo, hi = rank > 0, rank < world - 1
        rc = cb(None, FS_XCHG_HALO, ptr[0] if lo else None, ptr[1] if lo else None, ptr[2] if hi else None,
                ptr[3] if hi else None, n)
        ok = rc == 0
        if lo:
            ok &= bool(np.all(bufs[1] == 10 * (rank - 1) + 2))      # lower neighbour's send_hi
        if hi:
            ok &= bool(np.all(bufs[3] == 10 * (rank + 1) + 0))      # upper neighbour's send_lo
        val = np.array([rank + 1.5], np.float64)
        rc = cb(None, FS_XCHG_SUM_F64, val.ctypes.data_as(C.c_void_p), None, None, None, 8)
        ok &= rc == 0 and val[0] == sum(r + 1.5 for r in range(world))
        result[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    result = ctx.Array("i", [-1] * world)
    port = 29800 + (os.getpid() + world * 7) % 150
    procs = [ctx.Process(target=fn, args=(r, world, port, *args, result)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
    for pr in procs:
        if pr.is_alive():
            pr.terminate()
    assert list(result) == [1] * world, list(result)


@pytest.mark.parametrize("world,dims,steps,k", [(2, (16, 12, 20), 4, 10), (3, (12, 12, 27), 3, 7),
                                                (2, (10, 8, 52), 2, 20)])   # 26 planes per rank: paired passes
def test_slab_decomposition_replay_matches_single_domain(world, dims, steps, k):
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    _spawn(_slab_worker, world, dims, steps, k)


def test_host_transport_callback_between_processes():
    _spawn(_host_callback_worker, 2)
    _spawn(_host_callback_worker, 3)
