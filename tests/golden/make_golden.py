#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU oracle (oracle/fs_oracle.c,
built with gcc -O2 -ffp-contract=off).  CPU only; rerun with `python tests/golden/make_golden.py`.

The reference repository holds no vectors for this path and cannot be run, so these fixtures do
not pin the oracle to upstream; they guard the oracle against drift (compiler, flags, edits) and
give the GPU tests inputs/outputs that do not depend on building the oracle on the GPU box.

Files:
  plume16.npz  -- 16^3, K=20: all 7 fields after steps 1, 2, 10 (source moved inside the box)
  plume32.npz  -- 32^3, K=20: all 7 fields after steps 2 and 10
  plume20_confined.npz -- 20^3, K=12, vorticity confinement on (defined-behaviour variant of the
                          shipped VORT_CONFINE build): ux, uy, uz, rho after steps 3 and 8
  plume24_shipped.npz -- 24x32x24, the shipped configuration (MGPCG pressure solve + vorticity
                          confinement, 30 CG iterations at most): ux, uy, uz, rho, p and the CG
                          iteration count after steps 2 and 6
  plume64_stats.npz -- 64^3, K=20, 100 steps: per-field sum (f64) / L2 / max / argmax every 10
                       steps, the rho centre line rho[nz/2, :, nx/2] and 4096 fixed probe points
                       (index list stored in the file) at steps 50 and 100
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

NAMES = ["ux", "uy", "uz", "rho", "T", "p", "div"]


def small(n, steps_keep, sy, sr):
    p = oracle.default_params(n, n, n, jacobi_iters=20, source_y=sy, source_radius=sr)
    o = oracle.Oracle(p)
    out = {"n": n, "jacobi_iters": 20, "source_y": sy, "source_radius": sr}
    for s in range(1, max(steps_keep) + 1):
        o.step()
        if s in steps_keep:
            for f, name in enumerate(NAMES):
                out["s%d_%s" % (s, name)] = o.get(f)
    return out


def confined(n, steps_keep):
    kw = dict(jacobi_iters=12, source_y=6.0, source_radius=4.0)
    o = oracle.Oracle(oracle.default_params(n, n, n, vort_confine=1, **kw))
    out = {"n": n, **kw}
    for s in range(1, max(steps_keep) + 1):
        o.step()
        if s in steps_keep:
            for f in (0, 1, 2, 3):
                out["s%d_%s" % (s, NAMES[f])] = o.get(f)
    return out


def shipped(dims, steps_keep):
    kw = dict(jacobi_iters=30, source_y=8.0, source_radius=5.0)
    o = oracle.Oracle(oracle.default_params(*dims, vort_confine=1, pressure_solver=oracle.PRESSURE_MGPCG, **kw))
    out = {"dims": np.array(dims), **kw}
    for s in range(1, max(steps_keep) + 1):
        o.step()
        if s in steps_keep:
            for f in (0, 1, 2, 3, 5):
                out["s%d_%s" % (s, NAMES[f])] = o.get(f)
            out["s%d_cg_iters" % s] = o.cg_info()[0]
    return out


def stats64():
    n = 64
    o = oracle.Oracle(oracle.default_params(n, n, n, jacobi_iters=20))
    rng = np.random.default_rng(20241111)
    out = {"n": n, "jacobi_iters": 20}
    probes = {}
    for f, name in enumerate(NAMES[:6]):
        size = o.field(f).size
        probes[name] = np.sort(rng.choice(size, 4096, replace=False)).astype(np.int64)
        out["probe_idx_" + name] = probes[name]
    for s in range(1, 101):
        o.step()
        if s % 10 == 0:
            for f, name in enumerate(NAMES[:6]):
                a = o.field(f)
                out["s%d_%s_sum" % (s, name)] = np.float64(a.sum(dtype=np.float64))
                out["s%d_%s_l2" % (s, name)] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
                out["s%d_%s_max" % (s, name)] = np.float32(np.abs(a).max())
                out["s%d_%s_argmax" % (s, name)] = np.int64(np.abs(a).argmax())
        if s in (50, 100):
            out["s%d_rho_centerline" % s] = o.field(3)[n // 2, :, n // 2].copy()
            for f, name in enumerate(NAMES[:6]):
                out["s%d_probe_%s" % (s, name)] = o.field(f).reshape(-1)[probes[name]].copy()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "plume16.npz"), **small(16, (1, 2, 10), 5.0, 4.0))
    np.savez_compressed(os.path.join(HERE, "plume32.npz"), **small(32, (2, 10), 8.0, 5.0))
    np.savez_compressed(os.path.join(HERE, "plume20_confined.npz"), **confined(20, (3, 8)))
    np.savez_compressed(os.path.join(HERE, "plume24_shipped.npz"), **shipped((24, 32, 24), (2, 6)))
    np.savez_compressed(os.path.join(HERE, "plume64_stats.npz"), **stats64())
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
