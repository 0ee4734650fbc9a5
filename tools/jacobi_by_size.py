#!/usr/bin/env python3
"""Rate of the blocked Jacobi passes on grids whose three pressure fields do or do not fit the
256 MiB Infinity Cache (512 x 512 x 64 = 192 MiB does, 512^3 = 1.5 GiB does not): cell-sweeps per
second, device time per fused pass and the variant the creation-time timing picked.

    python tools/jacobi_by_size.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsolver_amd as fs
for dims in ((512, 512, 512), (512, 512, 128), (512, 512, 64), (512, 512, 32), (256, 256, 256)):
    s = fs.Solver(fs.default_params(*dims, jacobi_iters=80)).Initialize()
    s.step(3); s.sync()
    s.profile(True); s.profile_reset()
    s.step(10); s.sync()
    pr = s.profile_read()
    ms, launches = pr["jacobi"]
    cells = dims[0] * dims[1] * dims[2]
    print(dims, "jacobi %.3f ms/step, %.1f us/pass, %.3e cell-sweeps/s, config %s" % (ms / 10, 1e3 * ms / launches, cells * 80 * 10 / (ms * 1e-3), s.jacobi_config()), flush=True)
    s.close()
