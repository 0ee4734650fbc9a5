#!/usr/bin/env python3
"""Randomised parity sweep of the single-GPU solver against the CPU oracle.

    python tools/fuzz_parity.py [cases] [seed] [seconds] [share of medium-sized grids]

Every case draws a grid shape (odd extents, extents of 1, multiples of 4 and 8, thin and flat
grids), a scheme, a sweep count, optional vorticity confinement, the MGPCG solver where the shape
allows it, a random initial state and a random tile variant / store policy / pass depth of the pressure kernel,
steps both sides a few times and compares every field: bit for bit with the Jacobi solver, to 1e-5
plus equal iteration counts with MGPCG.  Prints one line per failing case and a summary; exit
status 1 if any case failed.  A complement to the fixed shapes of the test suite, not part of it.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


MEDIUM = 0.0    # share of medium-sized grids (fourth argument)


def draw_dims(rng, mgpcg):
    if mgpcg:
        return tuple(int(8 * rng.integers(1, 9)) for _ in range(3))
    kind = rng.integers(0, 5)
    if rng.random() < MEDIUM:      # several tiles in x and y, several z chunks, many workgroups
        return (int(4 * rng.integers(30, 100)), int(rng.integers(100, 330)), int(rng.integers(20, 130)))
    if kind == 0:
        return tuple(int(rng.integers(1, 40)) for _ in range(3))
    if kind == 1:      # rows the blocked pressure kernel takes (nx % 4 == 0), ragged elsewhere
        return (int(4 * rng.integers(2, 40)), int(rng.integers(1, 90)), int(rng.integers(1, 30)))
    if kind == 2:      # flat or thin
        d = [int(rng.integers(20, 140)), int(rng.integers(20, 140)), int(rng.integers(1, 4))]
        rng.shuffle(d)
        return tuple(d)
    if kind == 3:      # around the tile edges of the kernels (multiples of 48, 56, 64 +- 1)
        base = [48, 56, 64, 96, 112, 128]
        return (int(rng.choice(base) + rng.integers(-1, 2) * 4), int(rng.choice(base) + rng.integers(-1, 2)),
                int(rng.integers(2, 20)))
    return (int(4 * rng.integers(2, 24)), int(4 * rng.integers(2, 24)), int(4 * rng.integers(2, 24)))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    global MEDIUM
    MEDIUM = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    import fluidsolver_amd as fs
    import oracle
    from conftest import random_fields

    rng = np.random.default_rng(seed)
    failed, done, t_start = 0, 0, time.time()
    for case in range(cases):
        if time.time() - t_start > budget:
            break
        mgpcg = rng.random() < 0.25
        nx, ny, nz = draw_dims(rng, mgpcg)
        scheme = int(rng.integers(0, 2))
        k = int(rng.integers(0, 23)) if not mgpcg else 30
        confine = bool(rng.random() < 0.3)
        steps = int(rng.integers(1, 4))
        vel = float(rng.choice([0.3, 1.5, 4.0]))
        os.environ["FS_JACOBI_CFG"] = str(int(rng.integers(0, 4)))
        os.environ["FS_JACOBI_NT"] = str(int(rng.integers(0, 2)))
        os.environ["FS_JACOBI_TUNE"] = "0"
        # the six-sweep passes wherever the grid takes them, with short z chunks in half of those cases
        os.environ["FS_JACOBI_DEEP"] = str(int(rng.integers(0, 2)))
        os.environ["FS_JACOBI_DEEP_LZ"] = str(int(rng.choice([0, int(rng.integers(3, 40))])))
        os.environ["FS_JACOBI_DEEP_NT"] = str(int(rng.integers(0, 2)))
        kw = dict(jacobi_iters=k, scheme=scheme, source_y=min(10.0, ny / 3.0), source_radius=min(7.0, min(nx, ny, nz) / 4.0))
        gkw, okw = dict(kw), dict(kw)
        if confine:
            gkw["flags"] = fs.FS_FLAG_VORT_CONFINE
            okw["vort_confine"] = 1
        if mgpcg:
            gkw["pressure_solver"] = fs.FS_PRESSURE_MGPCG
            okw["pressure_solver"] = oracle.PRESSURE_MGPCG
        desc = "case %d: %dx%dx%d scheme %d k %d confine %d mgpcg %d steps %d vel %.1f cfg %s nt %s deep %s/%s/%s" % (
            case, nx, ny, nz, scheme, k, confine, mgpcg, steps, vel, os.environ["FS_JACOBI_CFG"], os.environ["FS_JACOBI_NT"],
            os.environ["FS_JACOBI_DEEP"], os.environ["FS_JACOBI_DEEP_LZ"], os.environ["FS_JACOBI_DEEP_NT"])
        try:
            g = fs.Solver(fs.default_params(nx, ny, nz, **gkw)).Initialize()
            o = oracle.Oracle(oracle.default_params(nx, ny, nz, **okw))
            st = random_fields(nx, ny, nz, seed=int(rng.integers(1, 1 << 30)), vel_scale=vel)
            for fid, arr in st.items():
                g.upload(fid, arr)
                o.set(fid, arr)
            bad = []
            for s in range(steps):
                g.step()
                o.step()
                if mgpcg and g.cg_info()[0] != o.cg_info()[0]:
                    bad.append("step %d: %d vs %d CG iterations" % (s, g.cg_info()[0], o.cg_info()[0]))
                for f in range(7):
                    a, b = g.field(f), o.get(f)
                    if mgpcg:
                        den = float(np.abs(b).max())
                        err = float(np.abs(a - b).max()) / den if den > 0 else float(np.abs(a).max())
                        if not err <= 1e-5:
                            bad.append("step %d field %d rel Linf %.2e" % (s, f, err))
                    elif not np.array_equal(a, b):
                        bad.append("step %d field %d: %d values differ" % (s, f, int((a != b).sum())))
                if bad:
                    break
            g.close()
            o.close()
        except Exception as e:   # noqa: BLE001
            bad = ["%s: %s" % (type(e).__name__, e)]
        done += 1
        if bad:
            failed += 1
            print("FAIL " + desc + " :: " + "; ".join(bad[:3]), flush=True)
        elif case % (20 if MEDIUM == 0 else 5) == 0:
            print("ok   " + desc, flush=True)
    print("%d cases, %d failed, %.0f s (seed %d)" % (done, failed, time.time() - t_start, seed), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
