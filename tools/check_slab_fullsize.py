#!/usr/bin/env python3
"""One-off parity check of a z-slab run at a BASELINE size against the CPU oracle.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 \
        --master-port P tools/check_slab_fullsize.py NX NY NZ STEPS SWEEPS [tolerance] [mgpcg] [confine] [random]

W ranks share whatever GPUs the box has (host-staged halos over gloo, the transport of the test
suite).  Rank 0 alone steps the single-domain oracle (minutes per step at 512 x 512 x 1024) and
broadcasts each of its fields; every rank compares its own planes bit for bit.  With a tolerance
the step runs in convergence mode (sum r^2 all-reduced every 8 sweeps) and the oracle repeats the
sweep count the slabs report.  `mgpcg` selects the MGPCG pressure solver (SWEEPS is then its
iteration limit), `confine` adds vorticity confinement; together they are the configuration the
reference ships.  `random` starts from a seeded random state (velocities of about a cell per step,
random density, temperature and pressure) instead of the zero state, so that every plane of every
field holds data.  Too slow for the test suite; its output is kept under profiles/.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(1100, exit=True)
    import torch
    import torch.distributed as dist
    import fluidsolver_amd as fs
    from fluidsolver_amd.slab import SlabSolver

    nx, ny, nz, steps, k = (int(v) for v in sys.argv[1:6])
    words = [w for w in sys.argv[6:] if w in ("mgpcg", "confine", "random")]
    rest = [w for w in sys.argv[6:] if w not in words]
    tol = float(rest[0]) if rest else 0.0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(jacobi_iters=k)
    okw = dict(jacobi_iters=k)
    if tol > 0:
        kw.update(tolerance=tol, check_every=8, max_sweeps=k)
    if "mgpcg" in words:
        kw["pressure_solver"] = fs.FS_PRESSURE_MGPCG
        okw["pressure_solver"] = 1
    if "confine" in words:
        kw["flags"] = fs.FS_FLAG_VORT_CONFINE
        okw["vort_confine"] = 1
    s = SlabSolver(fs.default_params(nx, ny, nz, device=dev, **kw), rank, world, transport="host").Initialize()
    o = None
    if rank == 0:
        import oracle
        o = oracle.Oracle(oracle.default_params(nx, ny, nz, **okw))
    ok = True
    if "random" in words:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import random_fields
        st = random_fields(nx, ny, nz, seed=77, vel_scale=1.5)     # the same on every rank
        for fid, arr in st.items():
            s.upload_global(fid, arr)
            if rank == 0:
                o.set(fid, arr)
        del st
    if rank == 0:
        print("passes of %d sweeps" % s.jacobi_config()["sweeps_per_pass"], flush=True)
    for step in range(steps):
        t0 = time.perf_counter()
        s.step()
        s.sync()
        t_gpu = time.perf_counter() - t0
        sweeps = s.last_sweeps
        t0 = time.perf_counter()
        if rank == 0:
            if tol > 0:
                import oracle
                o.stage("source"); o.stage("force"); o.stage("advect")
                d = oracle.divergence(o.get(0), o.get(1), o.get(2))
                pp = o.get(5)
                for _ in range(sweeps):
                    pp, _r = oracle.jacobi(pp, d, want_r=False)
                o.set(5, pp); o.set(6, d)
                o.stage("gradient")
            else:
                o.step()
        t_cpu = time.perf_counter() - t0
        bad = 0
        for fid in range(6):
            shape = fs.field_shape(fid, nx, ny, nz)
            want = torch.from_numpy(np.ascontiguousarray(o.get(fid))) if rank == 0 else torch.empty(shape, dtype=torch.float32)
            dist.broadcast(want, src=0)
            got = s.field(fid)
            mine = want.numpy()[s.z0:s.z0 + got.shape[0]]
            n = int((got != mine).sum())
            if n and "mgpcg" in words:
                # the dot products are summed per rank and then over the ranks: a different order
                # from the single domain's, so the bar is the one of the MGPCG tests (1e-5)
                err = float(np.abs(got - mine).max()) / max(float(np.abs(want.numpy()).max()), 1e-30)
                print("rank %d step %d field %s: %d of %d values differ, rel Linf %.2e" % (
                    rank, step + 1, fs.FIELD_NAMES[fid], n, got.size, err), flush=True)
                if err > 1e-5:
                    bad += n
            elif n:
                print("rank %d step %d field %s: %d of %d values differ" % (rank, step + 1, fs.FIELD_NAMES[fid], n, got.size),
                      flush=True)
                bad += n
        flag = torch.tensor([bad])
        dist.all_reduce(flag)
        ok = ok and int(flag.item()) == 0
        if "mgpcg" in words:
            it = torch.tensor([o.cg_info()[0] if rank == 0 else 0])
            dist.broadcast(it, src=0)
            if int(it.item()) != s.cg_info()[0]:
                print("rank %d step %d: %d CG iterations, the oracle %d" % (rank, step + 1, s.cg_info()[0], int(it.item())),
                      flush=True)
                ok = False
        if rank == 0:
            print("step %d: %d sweeps, slabs %.2f s (first step includes creation effects), oracle %.1f s, "
                  "mismatching values over all ranks and fields: %d" % (step + 1, sweeps, t_gpu, t_cpu, int(flag.item())),
                  flush=True)
    s.close()
    dist.destroy_process_group()
    if rank == 0:
        print("%dx%dx%d on %d ranks, %d steps, %s%s: %s" % (nx, ny, nz, world, steps,
              ("convergence mode, tolerance %g" % tol) if tol > 0 else
              ("MGPCG, at most %d iterations" % k) if "mgpcg" in words else "%d sweeps" % k,
              (", vorticity confinement" if "confine" in words else "") + (", random initial state" if "random" in words else ""),
              "BIT-IDENTICAL to the oracle" if ok else "MISMATCH"), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
