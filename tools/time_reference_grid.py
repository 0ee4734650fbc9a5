#!/usr/bin/env python3
"""Step time on the reference's own 64 x 128 x 64 grid: 30 Jacobi sweeps, and the configuration the
reference ships (MGPCG + vorticity confinement).  1000 timed steps after 100 of warm-up each.

    python tools/time_reference_grid.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import fluidsolver_amd as fs
    for name, kw in (("30 Jacobi sweeps", dict(jacobi_iters=30)),
                     ("shipped configuration", dict(jacobi_iters=30, flags=fs.FS_FLAG_VORT_CONFINE,
                                                    pressure_solver=fs.FS_PRESSURE_MGPCG))):
        s = fs.Solver(fs.default_params(64, 128, 64, **kw)).Initialize()
        s.step(100)
        s.sync()
        t = time.perf_counter()
        s.step(1000)
        s.sync()
        print("64x128x64, %s: %.3f ms/step" % (name, (time.perf_counter() - t)))   # 1000 steps: seconds = ms/step
        s.close()


if __name__ == "__main__":
    main()
