#!/usr/bin/env python3
"""Randomised parity sweep of the z-slab decomposition against the single-domain CPU oracle.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 \
        --master-port P tools/fuzz_slab.py [cases] [seed] [seconds] [jacobi|mgpcg|medium|all]

All ranks (sharing whatever GPUs the box has, host-staged halos over gloo) draw the same cases from
the same seed: grid shape (ragged slabs, thin slabs of 5 planes, slabs thick enough for the paired
pressure passes), scheme, sweep count, tile variant, vorticity confinement and the MGPCG solver
where the decomposition allows them, a random initial state.  Every rank compares its own planes of
every field with the oracle's after each step: bit for bit with the Jacobi solver, to 1e-5 plus
equal iteration counts with MGPCG.  A configuration the library refuses (slabs too thin for the
confinement's reach, ...) must be refused on every rank and is skipped.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import faulthandler
    faulthandler.dump_traceback_later(1100, exit=True)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    which = sys.argv[4] if len(sys.argv) > 4 else "all"
    # every rank steps its own copy of the oracle: keep the ranks' OpenMP teams within the cores
    # (oversubscribed teams spin in the barriers of the small multigrid loops for seconds)
    os.environ["OMP_NUM_THREADS"] = str(max(1, 8 // int(os.environ.get("WORLD_SIZE", "1"))))
    os.environ["OMP_WAIT_POLICY"] = "passive"
    import torch
    import torch.distributed as dist
    import fluidsolver_amd as fs
    from fluidsolver_amd.slab import SlabSolver
    import oracle
    from conftest import random_fields

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    failed = done = skipped = 0
    t_start = time.time()
    for case in range(cases):
        stop = torch.tensor([1 if time.time() - t_start > budget else 0])
        dist.all_reduce(stop, op=dist.ReduceOp.MAX)
        if int(stop.item()):
            break
        mgpcg = rng.random() < 0.3
        if which != "all":
            mgpcg = which == "mgpcg"
        mg_level = 4
        if mgpcg:
            # every depth of the V-cycle, coarsest slabs of odd and even thickness, the blocked
            # smoother on every level it can take or on none, every ghost depth of the windows
            mg_level = int(rng.integers(1, 5))
            m = 1 << (mg_level - 1)
            nx, ny = (int(m * rng.integers(1, max(2, 56 // m))) for _ in range(2))
            nz = int(m * world * rng.integers(1, max(2, 40 // m)))
            os.environ["FS_MG_TB_MIN"] = str(int(rng.choice([0, 128 * 128])))
            os.environ["FS_MG_SLAB_GHOST"] = str(int(rng.integers(1, 9)))
        else:
            nx = int(rng.choice([int(rng.integers(1, 40)), int(4 * rng.integers(2, 30))]))
            ny = int(rng.integers(1, 70))
            nz = int(world * 5 + rng.integers(0, 30 * world))
            if which == "medium":     # several tiles per plane, slabs thick enough for the paired pressure passes
                nx, ny = int(4 * rng.integers(20, 60)), int(rng.integers(60, 200))
                nz = int(world * 20 + rng.integers(0, 30 * world))
        scheme = int(rng.integers(0, 2))
        k = 30 if mgpcg else int(rng.integers(0, 26))
        confine = bool(rng.random() < 0.3)
        steps = int(rng.integers(1, 4))
        vel = float(rng.choice([0.3, 1.5]))     # (faster fields leave the ghost planes: FS_ERR_HALO, tested elsewhere)
        os.environ["FS_JACOBI_CFG"] = str(int(rng.integers(0, 4)))
        os.environ["FS_JACOBI_TUNE"] = "0"
        # six-sweep passes where the slabs can take them (17 planes, 8 ghost planes, 14 rows)
        os.environ["FS_JACOBI_DEEP"] = str(int(rng.integers(0, 2)))
        state_seed = int(rng.integers(1, 1 << 30))
        kw = dict(jacobi_iters=k, scheme=scheme, source_y=min(10.0, ny / 3.0), source_radius=min(7.0, min(nx, ny, nz) / 4.0))
        gkw, okw = dict(kw, device=dev), dict(kw)
        if confine:
            gkw["flags"] = fs.FS_FLAG_VORT_CONFINE
            okw["vort_confine"] = 1
        if mgpcg:
            gkw.update(pressure_solver=fs.FS_PRESSURE_MGPCG, mg_level=mg_level)
            okw.update(pressure_solver=oracle.PRESSURE_MGPCG, mg_level=mg_level)
        desc = "case %d: %dx%dx%d on %d ranks, scheme %d k %d confine %d mgpcg %d steps %d vel %.1f cfg %s deep %s" % (
            case, nx, ny, nz, world, scheme, k, confine, mgpcg, steps, vel, os.environ["FS_JACOBI_CFG"],
            os.environ["FS_JACOBI_DEEP"])
        if mgpcg:
            desc += " levels %d tb_min %s ghost %s" % (mg_level, os.environ["FS_MG_TB_MIN"], os.environ["FS_MG_SLAB_GHOST"])
        bad = []
        s = o = None
        t_case = time.time()
        t_gpu = t_cpu = 0.0
        try:
            s = SlabSolver(fs.default_params(nx, ny, nz, **gkw), rank, world, transport="host").Initialize()
            created = 1
        except fs.FsError:
            created = 0
        flags = [None] * world
        dist.all_gather_object(flags, created)
        if not all(flags):
            if any(flags):
                bad.append("refused on some ranks only: %r" % (flags,))
                if s is not None:
                    s.close()
            else:
                skipped += 1
                continue
        if not bad:
            o = oracle.Oracle(oracle.default_params(nx, ny, nz, **okw))
            st = random_fields(nx, ny, nz, seed=state_seed, vel_scale=vel)
            for fid, arr in st.items():
                s.upload_global(fid, arr)
                o.set(fid, arr)
            try:
                for step in range(steps):
                    t0 = time.time()
                    s.step()
                    s.sync()
                    t1 = time.time()
                    o.step()
                    t_gpu += t1 - t0
                    t_cpu += time.time() - t1
                    if mgpcg and s.cg_info()[0] != o.cg_info()[0]:
                        bad.append("step %d: %d vs %d CG iterations" % (step, s.cg_info()[0], o.cg_info()[0]))
                    for f in range(6):
                        a = s.field(f)
                        b = o.get(f)[s.z0:s.z0 + a.shape[0]]
                        if mgpcg:
                            den = float(np.abs(o.get(f)).max())
                            err = float(np.abs(a - b).max()) / den if den > 0 else float(np.abs(a).max())
                            if not err <= 1e-5:
                                bad.append("step %d field %d rel Linf %.2e" % (step, f, err))
                        elif not np.array_equal(a, b):
                            bad.append("step %d field %d: %d values differ" % (step, f, int((a != b).sum())))
                    allbad = [None] * world
                    dist.all_gather_object(allbad, len(bad))
                    if any(allbad):
                        break
            except fs.FsError as e:
                bad.append("FsError %s" % e)
                allbad = [None] * world
                dist.all_gather_object(allbad, len(bad))
            s.close()
            o.close()
        done += 1
        nbad = [None] * world
        dist.all_gather_object(nbad, bad[:3])
        if any(nbad):
            failed += 1
            if rank == 0:
                print("FAIL " + desc + " :: " + "; ".join("rank %d: %s" % (r, ", ".join(b)) for r, b in enumerate(nbad) if b),
                      flush=True)
        elif rank == 0 and case % 5 == 0:
            print("ok   " + desc + "  [%.1f s: slabs %.2f, oracle %.2f]" % (time.time() - t_case, t_gpu, t_cpu), flush=True)
    if rank == 0:
        print("%d cases on %d ranks, %d failed, %d refused by every rank and skipped, %.0f s (seed %d)" % (
            done, world, failed, skipped, time.time() - t_start, seed), flush=True)
    dist.destroy_process_group()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
