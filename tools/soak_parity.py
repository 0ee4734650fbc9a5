#!/usr/bin/env python3
"""Long runs of the single-GPU solver beside the CPU oracle at 64^3, compared every 100 steps.

    python tools/soak_parity.py [steps of the Jacobi configuration] [steps of the shipped one]

The Jacobi configuration (20 sweeps) and the configuration the reference ships (MGPCG + vorticity
confinement) step from the zero state with the sphere source, graph replay on; every field must
equal the oracle's bit for bit at every check.  Exit status 1 on the first difference.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import fluidsolver_amd as fs
    import oracle
    plan = [("jacobi", int(sys.argv[1]) if len(sys.argv) > 1 else 1000, dict(jacobi_iters=20), dict(jacobi_iters=20)),
            ("shipped", int(sys.argv[2]) if len(sys.argv) > 2 else 300,
             dict(jacobi_iters=30, flags=fs.FS_FLAG_VORT_CONFINE, pressure_solver=fs.FS_PRESSURE_MGPCG),
             dict(jacobi_iters=30, vort_confine=1, pressure_solver=oracle.PRESSURE_MGPCG))]
    for name, steps, gkw, okw in plan:
        g = fs.Solver(fs.default_params(64, 64, 64, **gkw)).Initialize()
        o = oracle.Oracle(oracle.default_params(64, 64, 64, **okw))
        t = time.time()
        for s in range(steps):
            g.step()
            o.step()
            if (s + 1) % 100 == 0 or s + 1 == steps:
                bad = [f for f in range(6) if not np.array_equal(g.field(f), o.get(f))]
                print("%s step %d: fields differing %r, %.0f s" % (name, s + 1, bad, time.time() - t), flush=True)
                if bad:
                    sys.exit(1)
        g.close()
        o.close()


if __name__ == "__main__":
    main()
