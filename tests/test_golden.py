"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from
the CPU oracle): the oracle must still reproduce them (CPU test), and the HIP path must match
them bit-for-bit / within the north-star tolerance (GPU tests) without any oracle on the box."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["ux", "uy", "uz", "rho", "T", "p", "div"]
REL_TOL = 1e-4   # north-star: max|a - b| / max|b| on density and velocity after 100 steps


def load(name):
    return np.load(os.path.join(HERE, "golden", name), allow_pickle=False)


def check_small(make, get, step_fn, fixture, steps):
    g = load(fixture)
    n = int(g["n"])
    s = make(n, int(g["jacobi_iters"]), float(g["source_y"]), float(g["source_radius"]))
    done = 0
    for st in steps:
        step_fn(s, st - done)
        done = st
        for f, name in enumerate(NAMES):
            want = g["s%d_%s" % (st, name)]
            got = get(s, f)
            assert np.array_equal(got, want), "%s step %d field %s differs" % (fixture, st, name)


def test_oracle_reproduces_small_fixtures(oracle):
    def make(n, k, sy, sr):
        return oracle.Oracle(oracle.default_params(n, n, n, jacobi_iters=k, source_y=sy, source_radius=sr))
    check_small(make, lambda s, f: s.get(f), lambda s, k: s.step(k), "plume16.npz", (1, 2, 10))
    check_small(make, lambda s, f: s.get(f), lambda s, k: s.step(k), "plume32.npz", (2, 10))


def test_oracle_reproduces_64cube_statistics(oracle):
    g = load("plume64_stats.npz")
    o = oracle.Oracle(oracle.default_params(64, 64, 64, jacobi_iters=20))
    o.step(50)
    for f, name in enumerate(NAMES[:6]):
        assert np.array_equal(o.field(f).reshape(-1)[g["probe_idx_" + name]], g["s50_probe_" + name])
        assert float(np.abs(o.field(f)).max()) == float(g["s50_%s_max" % name])
    assert np.array_equal(o.field(3)[32, :, 32], g["s50_rho_centerline"])


def check_confined(make, get, step_fn):
    g = load("plume20_confined.npz")
    n = int(g["n"])
    s = make(n, int(g["jacobi_iters"]), float(g["source_y"]), float(g["source_radius"]))
    done = 0
    for st in (3, 8):
        step_fn(s, st - done)
        done = st
        for f in (0, 1, 2, 3):
            assert np.array_equal(get(s, f), g["s%d_%s" % (st, NAMES[f])]), "confined step %d %s" % (st, NAMES[f])


def test_oracle_reproduces_confined_fixture(oracle):
    check_confined(lambda n, k, sy, sr: oracle.Oracle(oracle.default_params(
        n, n, n, jacobi_iters=k, source_y=sy, source_radius=sr, vort_confine=1)),
        lambda s, f: s.get(f), lambda s, k: s.step(k))


@pytest.mark.gpu
def test_gpu_matches_confined_fixture(fs):
    check_confined(lambda n, k, sy, sr: fs.Solver(fs.default_params(
        n, n, n, jacobi_iters=k, source_y=sy, source_radius=sr, flags=fs.FS_FLAG_VORT_CONFINE)).Initialize(),
        lambda s, f: s.field(f), lambda s, k: s.step(k))


def check_shipped(make, get, step_fn, iters_fn, exact):
    g = load("plume24_shipped.npz")
    dims = [int(v) for v in g["dims"]]
    s = make(dims, int(g["jacobi_iters"]), float(g["source_y"]), float(g["source_radius"]))
    done = 0
    for st in (2, 6):
        step_fn(s, st - done)
        done = st
        assert iters_fn(s) == int(g["s%d_cg_iters" % st])
        for f in (0, 1, 2, 3, 5):
            a, b = get(s, f), g["s%d_%s" % (st, NAMES[f])]
            if exact:
                assert np.array_equal(a, b), "shipped step %d %s" % (st, NAMES[f])
            else:   # the dot products are summed in another order: equal unless a sum sits on a rounding boundary
                assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-30), "shipped step %d %s" % (st, NAMES[f])


def test_oracle_reproduces_shipped_fixture(oracle):
    check_shipped(lambda d, k, sy, sr: oracle.Oracle(oracle.default_params(
        *d, jacobi_iters=k, source_y=sy, source_radius=sr, vort_confine=1, pressure_solver=oracle.PRESSURE_MGPCG)),
        lambda s, f: s.get(f), lambda s, k: s.step(k), lambda s: s.cg_info()[0], True)


@pytest.mark.gpu
def test_gpu_matches_shipped_fixture(fs):
    check_shipped(lambda d, k, sy, sr: fs.Solver(fs.default_params(
        *d, jacobi_iters=k, source_y=sy, source_radius=sr, flags=fs.FS_FLAG_VORT_CONFINE,
        pressure_solver=fs.FS_PRESSURE_MGPCG)).Initialize(),
        lambda s, f: s.field(f), lambda s, k: s.step(k), lambda s: s.cg_info()[0], False)


@pytest.mark.gpu
def test_gpu_matches_small_fixtures(fs):
    def make(n, k, sy, sr):
        return fs.Solver(fs.default_params(n, n, n, jacobi_iters=k, source_y=sy, source_radius=sr)).Initialize()
    check_small(make, lambda s, f: s.field(f), lambda s, k: s.step(k), "plume16.npz", (1, 2, 10))
    check_small(make, lambda s, f: s.field(f), lambda s, k: s.step(k), "plume32.npz", (2, 10))


@pytest.mark.gpu
def test_gpu_matches_64cube_100_step_statistics(fs):
    """BASELINE config 1 (64^3, 100 steps, 20 sweeps) against the committed statistics."""
    g = load("plume64_stats.npz")
    s = fs.Solver(fs.default_params(64, 64, 64, jacobi_iters=20)).Initialize()
    for st in range(10, 101, 10):
        s.step(10)
        for f, name in enumerate(NAMES[:6]):
            a = s.field(f)
            assert float(np.abs(a).max()) == float(g["s%d_%s_max" % (st, name)])
            assert int(np.abs(a).argmax()) == int(g["s%d_%s_argmax" % (st, name)])
            assert float(a.sum(dtype=np.float64)) == pytest.approx(float(g["s%d_%s_sum" % (st, name)]), rel=1e-12, abs=1e-12)
        if st in (50, 100):
            assert np.array_equal(s.field(3)[32, :, 32], g["s%d_rho_centerline" % st])
            for f, name in enumerate(NAMES[:6]):
                got = s.field(f).reshape(-1)[g["probe_idx_" + name]]
                want = g["s%d_probe_%s" % (st, name)]
                denom = float(np.abs(want).max()) or 1.0
                assert float(np.abs(got - want).max()) / denom <= REL_TOL
                assert np.array_equal(got, want)
    s.close()
