"""Worker of the multi-rank slab tests: run under torch.distributed.run (one process per rank).

    python -m torch.distributed.run --nproc-per-node W tests/slab_worker.py MODE NX NY NZ STEPS K [flags]

Every rank drives its z-slab through the C ABI and the ranks' planes are gathered and compared
bit-for-bit with the single-domain CPU oracle.  MODE = host (halos through host buffers and
torch.distributed; ranks may share one GPU) or rccl (one GPU per rank).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def self_peer(fs, nx, ny, nz, steps, kw):
    """The middle slab of three, alone on this GPU, trading its boundary planes with itself over
    RCCL (FS_FLAG_TEST_SELF_PEER): no oracle for that, but the run with the transfers on the
    communication stream (two events per exchange) must equal the run with everything on the
    compute stream, bit for bit, and neither may hang."""
    import ctypes as C
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)
    rng = np.random.default_rng(11)
    init = None
    runs = []
    for streams, extra_flags in (("1", 0), ("0", 0)):
        os.environ["FS_XCHG_STREAMS"] = streams
        p = fs.default_params(nx, ny, 3 * nz, source_z=1.5 * nz, flags=fs.FS_FLAG_TEST_SELF_PEER | extra_flags, **kw)
        s = slabmod.SlabSolver(p, 1, 3, transport="rccl")
        uid = (C.c_char * slabmod.FS_UNIQUE_ID_BYTES)()
        s._check(L.fs_comm_unique_id(uid))
        h = C.c_void_p()
        s._check(L.fs_create_slab(C.byref(p), 1, 3, uid, slabmod._EXCHANGE_FN(0), None, C.byref(h)))
        s._h = h
        if init is None:
            init = {f: (rng.standard_normal(s.local_shape(f)) * (0.8 if f < 3 else 0.3)).astype(np.float32)
                    for f in range(5)}
        for f, a in init.items():
            s.upload(f, a)
        s.step(steps)
        s.sync()
        runs.append([s.field(f) for f in range(6)])
        s.close()
    ok = all(np.array_equal(a, b) for a, b in zip(*runs)) and all(np.isfinite(a).all() for a in runs[0])
    moved = not np.array_equal(runs[0][0], init[0])
    # convergence mode on the same slab: the residual norm goes through ncclAllReduce
    del os.environ["FS_XCHG_STREAMS"]
    p = fs.default_params(nx, ny, 3 * nz, source_z=1.5 * nz, tolerance=1e-2, check_every=8, max_sweeps=64,
                          flags=fs.FS_FLAG_TEST_SELF_PEER,
                          **{a: b for a, b in kw.items() if a != "jacobi_iters"})
    s = slabmod.SlabSolver(p, 1, 3, transport="rccl")
    uid = (C.c_char * slabmod.FS_UNIQUE_ID_BYTES)()
    s._check(L.fs_comm_unique_id(uid))
    h = C.c_void_p()
    s._check(L.fs_create_slab(C.byref(p), 1, 3, uid, slabmod._EXCHANGE_FN(0), None, C.byref(h)))
    s._h = h
    for f, a in init.items():
        s.upload(f, a)
    s.step(2)
    res = s.residual(0)
    ok = ok and 8 <= s.last_sweeps <= 64 and np.isfinite(res) and (res < 1e-2 or s.last_sweeps == 64)
    s.close()
    print("SLAB_OK" if ok and moved else "SLAB_FAIL")
    sys.exit(0 if ok and moved else 1)


def self_peer_mgpcg(fs, nx, ny, nz, steps, kw):
    """FS_PRESSURE_MGPCG on the middle slab of three with itself as both neighbours: the run whose
    windows travel over RCCL send/recv and whose dot products go through ncclAllReduce must equal,
    bit for bit, the run on the host-staged transport with a loop-back exchange function."""
    import ctypes as C
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)

    def fn(ctx, op, send_lo, recv_lo, send_hi, recv_hi, nbytes):
        if op == slabmod.FS_XCHG_HALO:
            # RCCL pairs the sends and receives of one peer in the order they were posted, so with
            # itself as both neighbours the planes sent downwards come back as the planes received
            # from below: the loop-back does the same (not a periodic domain, but the same system)
            C.memmove(recv_lo, send_lo, nbytes)
            C.memmove(recv_hi, send_hi, nbytes)
        return 0          # FS_XCHG_SUM_F64 over one rank: the value stays

    cb = slabmod._EXCHANGE_FN(fn)
    rng = np.random.default_rng(13)
    init, runs, iters = None, [], []
    for transport in ("rccl", "host"):
        p = fs.default_params(nx, ny, 3 * nz, source_z=1.5 * nz, pressure_solver=fs.FS_PRESSURE_MGPCG,
                              flags=fs.FS_FLAG_TEST_SELF_PEER | fs.FS_FLAG_VORT_CONFINE, **kw)
        s = slabmod.SlabSolver(p, 1, 3, transport=transport)
        h = C.c_void_p()
        if transport == "rccl":
            uid = (C.c_char * slabmod.FS_UNIQUE_ID_BYTES)()
            s._check(L.fs_comm_unique_id(uid))
            s._check(L.fs_create_slab(C.byref(p), 1, 3, uid, slabmod._EXCHANGE_FN(0), None, C.byref(h)))
        else:
            s._check(L.fs_create_slab(C.byref(p), 1, 3, None, cb, None, C.byref(h)))
        s._h = h
        if init is None:
            init = {f: (rng.standard_normal(s.local_shape(f)) * (0.8 if f < 3 else 0.3)).astype(np.float32)
                    for f in range(5)}
        for f, a in init.items():
            s.upload(f, a)
        s.step(steps)
        s.sync()
        iters.append(s.cg_info()[0])
        runs.append([s.field(f) for f in range(6)])
        s.close()
    ok = all(np.array_equal(a, b) for a, b in zip(*runs)) and all(np.isfinite(a).all() for a in runs[0])
    ok = ok and iters[0] == iters[1] and iters[0] >= 1 and not np.array_equal(runs[0][0], init[0])
    print("SLAB_OK" if ok else "SLAB_FAIL (CG iterations %r)" % iters)
    sys.exit(0 if ok else 1)


def self_peer_host(fs, nx, ny, nz, steps, kw):
    """The same middle slab on the host-staged transport, still one process: the exchange function
    hands the planes sent upwards back as the planes received from below and vice versa.  The
    two-stream ordering must equal the one-stream ordering bit for bit."""
    import ctypes as C
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)

    def fn(ctx, op, send_lo, recv_lo, send_hi, recv_hi, nbytes):
        if op == slabmod.FS_XCHG_HALO:
            C.memmove(recv_lo, send_hi, nbytes)
            C.memmove(recv_hi, send_lo, nbytes)
        return 0

    cb = slabmod._EXCHANGE_FN(fn)
    rng = np.random.default_rng(12)
    init, runs = None, []
    for streams in ("1", "0"):
        os.environ["FS_XCHG_STREAMS"] = streams
        p = fs.default_params(nx, ny, 3 * nz, source_z=1.5 * nz, flags=fs.FS_FLAG_TEST_SELF_PEER, **kw)
        s = slabmod.SlabSolver(p, 1, 3, transport="host")
        h = C.c_void_p()
        s._check(L.fs_create_slab(C.byref(p), 1, 3, None, cb, None, C.byref(h)))
        s._h = h
        if init is None:
            init = {f: (rng.standard_normal(s.local_shape(f)) * (0.8 if f < 3 else 0.3)).astype(np.float32)
                    for f in range(5)}
        for f, a in init.items():
            s.upload(f, a)
        s.step(steps)
        s.sync()
        runs.append([s.field(f) for f in range(6)])
        s.close()
    ok = all(np.array_equal(a, b) for a, b in zip(*runs)) and all(np.isfinite(a).all() for a in runs[0])
    moved = not np.array_equal(runs[0][0], init[0])
    print("SLAB_OK" if ok and moved else "SLAB_FAIL")
    sys.exit(0 if ok and moved else 1)


def main():
    # a stuck worker prints the Python stacks of its threads and exits instead of waiting for the
    # parent's time limit
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("FS_WORKER_DUMP_AFTER", "200")), exit=True)
    import torch
    import torch.distributed as dist
    import fluidsolver_amd as fs
    from fluidsolver_amd.slab import SlabSolver
    import oracle
    from conftest import random_fields

    mode = sys.argv[1]
    nx, ny, nz, steps, k = (int(v) for v in sys.argv[2:7])
    extra = sys.argv[7:]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(jacobi_iters=k, source_y=min(10.0, ny / 3.0), source_radius=min(7.0, min(nx, ny, nz) / 4.0))
    if mode == "selfpeer":
        return self_peer(fs, nx, ny, nz, steps, kw)
    if mode == "selfpeer_host":
        return self_peer_host(fs, nx, ny, nz, steps, kw)
    if mode == "selfpeer_mgpcg":
        return self_peer_mgpcg(fs, nx, ny, nz, steps, kw)
    if "naive" in extra:
        kw["flags"] = fs.FS_FLAG_JACOBI_NAIVE
    if "converge" in extra:
        kw.update(tolerance=1e-3, check_every=8, max_sweeps=400)
    if "confine" in extra:
        kw["flags"] = kw.get("flags", 0) | fs.FS_FLAG_VORT_CONFINE
    if "mgpcg" in extra:
        kw["pressure_solver"] = fs.FS_PRESSURE_MGPCG
    for e in extra:
        if e.startswith("mglevel="):
            kw["mg_level"] = int(e.split("=")[1])
    p = fs.default_params(nx, ny, nz, device=dev, **kw)
    s = SlabSolver(p, rank, world, transport=mode).Initialize()
    okw = {a: b for a, b in kw.items() if a not in ("flags", "tolerance", "check_every", "max_sweeps")}
    for word, want in (("deep", 6), ("four", 4)):
        # the caller expects a decomposition that runs six-sweep (four-sweep) passes: make sure it does
        if word in extra and s.jacobi_config()["sweeps_per_pass"] != want:
            print("rank %d: passes of %r sweeps, not %d" % (rank, s.jacobi_config()["sweeps_per_pass"], want))
            print("SLAB_FAIL")
            sys.exit(1)
    if "confine" in extra:
        okw["vort_confine"] = 1
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, **okw))
    ok = True
    if "random" in extra:
        st = random_fields(nx, ny, nz, seed=5, vel_scale=1.5)
        for fid, arr in st.items():
            s.upload_global(fid, arr)
            o.set(fid, arr)
    if "toofast" in extra:
        # a uniform uz whose back-trace crosses more planes per step than the slabs' ghost depth:
        # every rank with a neighbour below must report FS_ERR_HALO at the next sync, none may
        # return fields computed from planes it does not hold
        for fid, shape_z in ((fs.FS_UX, nz), (fs.FS_UY, nz), (fs.FS_UZ, nz + 1)):
            full = np.zeros((shape_z,) + fs.field_shape(fid, nx, ny, nz)[1:], np.float32)
            if fid == fs.FS_UZ:
                full[:] = 30.0
            s.upload_global(fid, full)
        code = 0
        try:
            s.step()
            s.sync()
        except fs.FsError as e:
            code = e.code
        codes = [None] * world
        dist.all_gather_object(codes, code)
        ok = all(c == (0 if r == 0 else -4) for r, c in enumerate(codes))
        if not ok:
            print("rank %d: error codes %r" % (rank, codes))
        steps = 0
    for step in range(steps):
        s.step()
        if "converge" in extra:
            sweeps = s.last_sweeps
            res = s.residual(0)
            o.stage("source"); o.stage("force"); o.stage("advect")
            d = oracle.divergence(o.get(0), o.get(1), o.get(2))
            pp = o.get(5)
            for _ in range(sweeps):
                pp, _r = oracle.jacobi(pp, d, want_r=False)
            o.set(5, pp); o.set(6, d)
            _, r = oracle.jacobi(pp, d)
            want = oracle.sum_sq(r)
            o.stage("gradient")
            if abs(res - want) > 1e-5 * max(want, 1e-30) or not (res < 1e-3 or sweeps == 400):
                print("rank %d step %d: residual %r vs oracle %r after %d sweeps" % (rank, step, res, want, sweeps))
                ok = False
        else:
            o.step()
        if "mgpcg" in extra and os.environ.get("FS_WORKER_VERBOSE"):
            print("rank %d step %d: CG %r oracle %r, rel Linf %s" % (rank, step + 1, s.cg_info(), o.cg_info(), [
                "%.1e" % (float(np.abs(s.gather(f) - o.get(f)).max()) / max(float(np.abs(o.get(f)).max()), 1e-30))
                for f in range(6)]), flush=True)
        if "mgpcg" in extra:
            # the dot products are sums over the ranks' partial sums: a different order from the
            # single domain's, so the CG scalars may differ in the last bit (same bar as the
            # single-GPU MGPCG tests: equal iteration counts, fields to 1e-5)
            gi, grtr = s.cg_info()
            oi, ortr = o.cg_info()
            if gi != oi or abs(grtr - ortr) > 1e-4 * max(ortr, 1e-12):
                print("rank %d step %d: %d CG iterations, rTr %r; oracle %d, %r" % (rank, step + 1, gi, grtr, oi, ortr))
                ok = False
        for fid in range(6):
            got = s.gather(fid)
            want = o.get(fid)
            if "mgpcg" in extra and got.shape == want.shape:
                err = float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)
                if err > 1e-5:
                    print("rank %d step %d field %d: rel Linf %.3e" % (rank, step + 1, fid, err))
                    ok = False
                continue
            if got.shape != want.shape or not np.array_equal(got, want):
                bad = np.argwhere(got != want) if got.shape == want.shape else []
                print("rank %d step %d field %d differs (%d points, first %s)"
                      % (rank, step + 1, fid, len(bad), bad[0] if len(bad) else got.shape))
                ok = False
        if not ok:
            break
    if steps:
        # asynchronous read-back on a slab: the pinned buffer holds the LOCAL planes only
        s.field_begin(fs.FS_UZ)
        got = s.field_end(fs.FS_UZ)
        if got.shape != s.local_shape(fs.FS_UZ) or not np.array_equal(got, s.field(fs.FS_UZ)):
            print("rank %d: asynchronous read-back of uz differs from the blocking one" % rank)
            ok = False
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    s.close()
    o.close()
    dist.destroy_process_group()
    if rank == 0:
        print("SLAB_OK" if int(flag.item()) == 1 else "SLAB_FAIL")
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
