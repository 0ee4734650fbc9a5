"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Bar: bit-exact (numpy ``==``, i.e. up to the sign of zero) for every field, because
both sides evaluate the reference expressions in binary32 without FMA contraction.
The north-star tolerance (L-inf relative error <= 1e-4 on density and velocity after
100 steps) is asserted as well, on the 64^3 / 100-step / K=20 configuration.
"""
import numpy as np
import pytest

from conftest import random_fields, smooth_velocity

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4   # north-star tolerance: max|gpu - oracle| / max|oracle|


def rel_linf(a, b):
    denom = float(np.abs(b).max())
    return float(np.abs(a - b).max()) / denom if denom > 0 else float(np.abs(a).max())


def make_pair(fs, oracle, nx, ny, nz, **kw):
    okw = {k: v for k, v in kw.items() if k not in ("flags",)}
    names = {"source_rho": "rho0", "source_temperature": "temperature0"}
    op = oracle.default_params(nx, ny, nz, **{names.get(k, k): v for k, v in okw.items()})
    gp = fs.default_params(nx, ny, nz, **kw)
    return fs.Solver(gp).Initialize(), oracle.Oracle(op)


def load_state(g, o, fields):
    for fid, arr in fields.items():
        g.upload(fid, arr)
        o.set(fid, arr)


def assert_all_equal(g, o, fields=range(7), what=""):
    for fid in fields:
        a, b = g.field(fid), o.get(fid)
        assert a.shape == b.shape
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            pytest.fail("%s field %d differs at %d of %d points, first %s gpu=%r oracle=%r, rel Linf %.3e"
                        % (what, fid, len(bad), a.size, bad[0], a[tuple(bad[0])], b[tuple(bad[0])],
                           rel_linf(a, b)))


SIZES = [(21, 13, 17), (16, 16, 16), (64, 5, 3), (1, 1, 1), (3, 70, 2), (65, 33, 9)]


@pytest.mark.parametrize("dims", SIZES)
def test_stage_kernels_random_state(fs, oracle, dims):
    """Every stage on its own, from the same seeded random state (ragged sizes included)."""
    nx, ny, nz = dims
    for stage in ("source", "force", "divergence", "gradient", "advect"):
        g, o = make_pair(fs, oracle, nx, ny, nz, source_x=nx / 2.0, source_y=min(10.0, ny / 2.0),
                         source_z=nz / 2.0, source_radius=min(7.0, max(nx, ny, nz) / 3.0))
        seed = 1000 * nx + 10 * ny + nz + len(stage)
        load_state(g, o, random_fields(nx, ny, nz, seed=seed, vel_scale=2.0))
        g.stage(stage)
        if stage == "divergence":
            d = oracle.divergence(o.get(0), o.get(1), o.get(2))
            assert np.array_equal(g.field(fs.FS_DIV), d)
        else:
            o.stage(stage)
            assert_all_equal(g, o, what="stage %s %s" % (stage, dims))
        g.close(); o.close()


@pytest.mark.parametrize("dims", SIZES)
@pytest.mark.parametrize("sweeps", [1, 2, 3, 7, 20])
def test_jacobi_random(fs, oracle, dims, sweeps):
    nx, ny, nz = dims
    g, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=sweeps)
    st = random_fields(nx, ny, nz, seed=7 + sweeps)
    load_state(g, o, st)
    g.stage("jacobi", sweeps)
    p, div = st[5], st[6]
    prev = p
    for _ in range(sweeps):
        prev = p
        p, _r = oracle.jacobi(p, div, want_r=False)
    assert np.array_equal(g.field(fs.FS_P), p)
    # residual of the current iterate, binary64 sum of binary32 r
    _, r = oracle.jacobi(p, div)
    want = oracle.sum_sq(r)
    got = g.residual(0)
    assert abs(got - want) <= 1e-5 * max(want, 1e-30)
    if sweeps % 2 == 1:
        # after an odd sweep count the previous iterate is still held, as in the reference
        _, r1 = oracle.jacobi(prev, div)
        want1 = oracle.sum_sq(r1)
        assert abs(g.residual(1) - want1) <= 1e-5 * max(want1, 1e-30)
    g.close(); o.close()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("dims", [(132, 70, 23), (64, 61, 9), (60, 124, 40), (8, 1, 5), (196, 130, 12)])
def test_jacobi_tile_variants(fs, oracle, dims, cfg, monkeypatch):
    """The tile variants of the blocked kernel (64- and 32-row tiles; output rows of 56 cells or
    of 48 cells on 64-byte boundaries) on grids with several tiles per direction, tile edges that coincide with walls, walls in mid-tile and single-tile grids;
    8 and 6 sweeps (two fused passes; a fused pass plus two single sweeps).  The 8-sweep run also
    forces the non-temporal store policy, which the timing at creation would only pick on grids
    larger than the Infinity Cache."""
    monkeypatch.setenv("FS_JACOBI_CFG", str(cfg))
    nx, ny, nz = dims
    for sweeps in (8, 6):
        monkeypatch.setenv("FS_JACOBI_NT", "1" if sweeps == 8 else "0")
        g, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=sweeps)
        st = random_fields(nx, ny, nz, seed=31 + sweeps + cfg)
        load_state(g, o, st)
        g.stage("jacobi", sweeps)
        p, div = st[5], st[6]
        for _ in range(sweeps):
            p, _r = oracle.jacobi(p, div, want_r=False)
        got = g.field(fs.FS_P)
        assert np.array_equal(got, p), "cfg %d dims %s: %d mismatches" % (cfg, dims, int((got != p).sum()))
        g.close(); o.close()


@pytest.mark.parametrize("dims", [(132, 70, 23), (64, 61, 13), (60, 124, 40), (8, 14, 12), (196, 130, 12), (48, 36, 30)])
def test_jacobi_six_sweep_passes(fs, oracle, dims, monkeypatch):
    """FS_JACOBI_DEEP=1: a single domain runs its sweeps in passes of six (two values per level in
    registers instead of three planes) wherever it can; 6, 12, 14 (6 + 4 + 4), 8 (4 + 4), 9 (6 + three
    single sweeps) and 20 (6 + 6 + 4 + 4) sweeps on grids with several tiles per direction, tile edges
    on walls and in mid-tile, fewer planes than a march needs to fill, and both store policies."""
    monkeypatch.setenv("FS_JACOBI_DEEP", "1")
    nx, ny, nz = dims
    for sweeps in (6, 12, 14, 8, 9, 20):
        monkeypatch.setenv("FS_JACOBI_DEEP_LZ", "0" if sweeps != 12 else "7")     # chunks shorter than their ramp
        monkeypatch.setenv("FS_JACOBI_DEEP_NT", "1" if sweeps == 14 else "0")
        g, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=sweeps)
        st = random_fields(nx, ny, nz, seed=131 + sweeps)
        load_state(g, o, st)
        g.stage("jacobi", sweeps)
        p, div = st[5], st[6]
        for _ in range(sweeps):
            p, _r = oracle.jacobi(p, div, want_r=False)
        got = g.field(fs.FS_P)
        assert np.array_equal(got, p), "dims %s, %d sweeps: %d mismatches" % (dims, sweeps, int((got != p).sum()))
        assert g.jacobi_config()["sweeps_per_pass"] == 6
        g.close(); o.close()


@pytest.mark.parametrize("brick", [0, 1])
@pytest.mark.parametrize("scheme", [0, 1])
@pytest.mark.parametrize("dims", [(24, 20, 18), (33, 9, 14)])
def test_advect_smooth_and_rough(fs, oracle, dims, scheme, brick, monkeypatch):
    """Advection with a smooth swirl that crosses every wall, and with rough random velocity
    (back-traces land far outside the box), RK4 and MacCormack, wide and narrow LDS bricks."""
    monkeypatch.setenv("FS_ADVECT_BRICK", str(brick))
    nx, ny, nz = dims
    for kind in ("smooth", "rough", "zero"):
        g, o = make_pair(fs, oracle, nx, ny, nz, scheme=scheme)
        st = random_fields(nx, ny, nz, seed=11, vel_scale=8.0)
        if kind == "smooth":
            st[0], st[1], st[2] = smooth_velocity(nx, ny, nz)
        elif kind == "zero":
            st[0][...] = 0; st[1][...] = 0; st[2][...] = 0
        load_state(g, o, st)
        g.stage("advect"); o.stage("advect")
        assert_all_equal(g, o, fields=range(5), what="advect %s scheme %d" % (kind, scheme))
        if kind == "zero":
            # zero velocity => the back-trace returns the sample point and the gather is exact
            assert np.array_equal(g.field(fs.FS_RHO), st[3])
            assert np.array_equal(g.field(fs.FS_T), st[4])
        g.close(); o.close()


def test_trilerp_quirk_window(fs, oracle):
    """Coordinates in (-1e-6, 0) and at exact negative integers follow the reference's
    int()/floor() fix-up (Solver.cu:58-82): drive them through a constant velocity."""
    nx, ny, nz = 12, 10, 8
    for ushift in (2.5, 5.0, 2.4999998, 1e-6 / 0.2, 4.9999995e-6, 7.5):
        g, o = make_pair(fs, oracle, nx, ny, nz, scheme=0)
        st = random_fields(nx, ny, nz, seed=3)
        st[0][...] = np.float32(ushift); st[1][...] = np.float32(ushift); st[2][...] = np.float32(ushift)
        load_state(g, o, st)
        g.stage("advect"); o.stage("advect")
        assert_all_equal(g, o, fields=range(5), what="quirk u=%r" % ushift)
        g.close(); o.close()


@pytest.mark.parametrize("n,steps,k", [(16, 10, 20), (32, 10, 20)])
def test_full_step_small(fs, oracle, n, steps, k):
    g, o = make_pair(fs, oracle, n, n, n, jacobi_iters=k, source_y=min(10.0, n / 3.0),
                     source_radius=min(7.0, n / 4.0))
    for s in range(steps):
        g.step(); o.step()
        if s in (0, 1, steps - 1):
            assert_all_equal(g, o, what="step %d" % (s + 1))
    g.close(); o.close()


@pytest.mark.parametrize("dims", [(4096, 4, 8), (8, 4096, 4), (4, 8, 4096), (4096, 1, 1)])
def test_largest_extents(fs, oracle, dims):
    """The largest extent the library accepts (4096) along each axis, on pencil-shaped grids: a
    random state through every stage, then full steps with the source on."""
    nx, ny, nz = dims
    kw = dict(jacobi_iters=9, source_x=nx / 2.0, source_y=ny / 2.0, source_z=nz / 2.0, source_radius=3.0)
    g, o = make_pair(fs, oracle, nx, ny, nz, **kw)
    load_state(g, o, random_fields(nx, ny, nz, seed=sum(dims), vel_scale=2.0))
    for stage in ("force", "advect", "project", "gradient"):
        g.stage(stage); o.stage(stage)
        assert_all_equal(g, o, fields=range(6), what="stage %s %s" % (stage, dims))
    g.step(3); o.step(3)
    assert_all_equal(g, o, what="steps %s" % (dims,))
    g.close(); o.close()


def test_default_shipped_grid(fs, oracle):
    """The reference's compiled-in grid 64x128x64 (Solver.h:9-11), 5 steps, K = 30."""
    g, o = make_pair(fs, oracle, 64, 128, 64)
    g.step(5); o.step(5)
    assert_all_equal(g, o, what="64x128x64")
    g.close(); o.close()


def test_config_64cube_100_steps(fs, oracle):
    """BASELINE config 1: 64^3 smoke plume, 100 steps, 20 Jacobi sweeps."""
    g, o = make_pair(fs, oracle, 64, 64, 64, jacobi_iters=20)
    g.step(100); o.step(100)
    for fid in (fs.FS_RHO, fs.FS_UX, fs.FS_UY, fs.FS_UZ):
        assert rel_linf(g.field(fid), o.get(fid)) <= REL_TOL
    assert_all_equal(g, o, what="64^3 x100")
    assert g.frame == 100
    g.close(); o.close()


def test_config_128cube_k40(fs, oracle):
    """BASELINE config 2: 128^3 smoke plume, 40 Jacobi sweeps (8 steps keep the oracle short)."""
    g, o = make_pair(fs, oracle, 128, 128, 128, jacobi_iters=40)
    g.step(8); o.step(8)
    for fid in (fs.FS_RHO, fs.FS_UX, fs.FS_UY, fs.FS_UZ):
        assert rel_linf(g.field(fid), o.get(fid)) <= REL_TOL
    assert_all_equal(g, o, what="128^3 x8")
    g.close(); o.close()


def test_convergence_mode(fs, oracle):
    """tolerance > 0: sweeps run in groups of check_every until sum r^2 < tolerance."""
    n = 24
    g, o = make_pair(fs, oracle, n, n, n, source_y=8.0, source_radius=5.0)
    g.close()
    gp = fs.default_params(n, n, n, source_y=8.0, source_radius=5.0, tolerance=1e-3, check_every=8,
                           max_sweeps=4000)
    g = fs.Solver(gp).Initialize()
    for _ in range(3):
        g.step()
        sweeps = g.last_sweeps
        assert sweeps % 8 == 0 and 0 < sweeps <= 4000
        assert g.residual(0) < 1e-3
        # oracle: same number of sweeps
        o.stage("source"); o.stage("force"); o.stage("advect")
        # the oracle's sweep count lives inside its C state, so sweep by hand
        d = oracle.divergence(o.get(0), o.get(1), o.get(2))
        p = o.get(5)
        for _s in range(sweeps):
            p, _r = oracle.jacobi(p, d, want_r=False)
        o.set(5, p); o.set(6, d)
        o.stage("gradient")
        assert_all_equal(g, o, what="convergence mode")
    g.close(); o.close()


@pytest.mark.parametrize("k", [8, 6, 5])
def test_graph_replay_matches_eager_launches(fs, oracle, k):
    """fs_step replays a captured graph per buffer assignment; FS_FLAG_NO_GRAPH launches eagerly.
    Both must give the same bits as the oracle, also when stage calls and uploads between steps
    change which buffer plays which role (k = 6 and 5 leave the pressure in the other buffer
    after a step, so consecutive steps alternate between assignments)."""
    nx, ny, nz = 24, 28, 20
    kw = dict(jacobi_iters=k, source_y=8.0, source_radius=5.0)
    g, o = make_pair(fs, oracle, nx, ny, nz, **kw)
    e, _o2 = make_pair(fs, oracle, nx, ny, nz, flags=fs.FS_FLAG_NO_GRAPH, **kw)
    _o2.close()
    for rnd in range(3):
        for _ in range(3):
            g.step(); e.step(); o.step()
        assert_all_equal(g, o, what="graph k=%d round %d" % (k, rnd))
        assert_all_equal(e, o, what="eager k=%d round %d" % (k, rnd))
        # break the period: one stage call that swaps buffers, and an upload
        for s in (g, e):
            s.stage("source"); s.stage("force"); s.stage("advect")
        o.stage("source"); o.stage("force"); o.stage("advect")
        st = random_fields(nx, ny, nz, seed=90 + rnd)
        for s in (g, e):
            s.upload(fs.FS_P, st[5])
        o.set(5, st[5])
    g.step(2); e.step(2); o.step(); o.step()
    assert_all_equal(g, o, what="graph k=%d final" % k)
    assert_all_equal(e, o, what="eager k=%d final" % k)
    # the graph path was really taken: at least one graph per buffer assignment met, none eagerly
    assert 1 <= g.step_graphs <= 8 and e.step_graphs == 0
    g.close(); e.close(); o.close()


@pytest.mark.parametrize("lattice", [1, 0])
@pytest.mark.parametrize("dims", [(16, 20, 12), (9, 8, 10), (6, 6, 6), (70, 9, 11), (67, 70, 9)])
def test_vorticity_confinement_stages(fs, oracle, dims, lattice, monkeypatch):
    """FS_FLAG_VORT_CONFINE: FindVort + ApplyVortConfinement from a random state, against the
    oracle's literal gathers (mismatched strides, guard from the launch dims, double sqrt); with
    the |omega| values shared through the four lattices (default) and with every face evaluating
    its own (FS_VORT_LATTICE=0)."""
    monkeypatch.setenv("FS_VORT_LATTICE", str(lattice))
    nx, ny, nz = dims
    g, o = make_pair(fs, oracle, nx, ny, nz, flags=fs.FS_FLAG_VORT_CONFINE)
    o.close()
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, vort_confine=1))
    st = random_fields(nx, ny, nz, seed=77, vel_scale=3.0)
    load_state(g, o, st)
    g.stage("vorticity"); o.stage("vorticity")
    g.stage("confine"); o.stage("confine")
    assert_all_equal(g, o, fields=range(3), what="confinement %s" % (dims,))
    # second application with the same vorticity on the changed velocity
    g.stage("confine"); o.stage("confine")
    assert_all_equal(g, o, fields=range(3), what="confinement twice %s" % (dims,))
    g.close(); o.close()


def test_vorticity_confinement_full_steps(fs, oracle):
    """The shipped reference build's step order (VORT_CONFINE 1) on a plume, graph replay on."""
    nx, ny, nz = 32, 40, 28
    kw = dict(jacobi_iters=12, source_y=8.0, source_radius=5.0)
    g, o = make_pair(fs, oracle, nx, ny, nz, flags=fs.FS_FLAG_VORT_CONFINE, **kw)
    o.close()
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, vort_confine=1, **kw))
    for s in range(12):
        g.step(); o.step()
        if s in (0, 5, 11):
            assert_all_equal(g, o, what="confined step %d" % (s + 1))
    g.close(); o.close()


@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_jacobi_division_on_hard_inputs(fs, oracle, cfg, monkeypatch):
    """The fused kernel divides by 6 through a binary64 multiply; the oracle uses the true
    division.  Inputs chosen to land on the hard cases: quotients that are exact ties of the
    subnormal grid (x = 3*(2m+1)*2^-149), fields that are subnormal throughout, and random bit
    patterns over the whole finite range (sums may overflow to infinity)."""
    monkeypatch.setenv("FS_JACOBI_CFG", str(cfg))
    nx, ny, nz = 72, 40, 12
    rng = np.random.default_rng(123 + cfg)
    shape = (nz, ny, nx)
    tiny = np.float32(2.0 ** -149)

    def run(p0, div, what):
        g, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=4)
        load_state(g, o, {5: p0, 6: div})
        g.stage("jacobi", 4)
        p = p0
        for _ in range(4):
            p, _r = oracle.jacobi(p, div, want_r=False)
        got = g.field(fs.FS_P)
        ok = (got == p) | (np.isnan(got) & np.isnan(p))
        assert ok.all(), "%s: %d mismatches, first gpu=%r oracle=%r" % (
            what, int((~ok).sum()), got[~ok][0], p[~ok][0])
        g.close(); o.close()

    with np.errstate(all="ignore"):
        m = rng.integers(0, 2 ** 20, shape)
        ties = (3 * (2 * m + 1)).astype(np.float32) * tiny * rng.choice(np.float32([-1, 1]), shape)
        run(np.zeros(shape, np.float32), -ties, "exact subnormal ties")
        sub = rng.integers(-2 ** 22, 2 ** 22, shape).astype(np.float32) * tiny
        run(sub, rng.integers(-2 ** 23, 2 ** 23, shape).astype(np.float32) * tiny, "subnormal fields")
        bits = rng.integers(0, 2 ** 32, shape, dtype=np.uint64).astype(np.uint32)
        bits[(bits >> 23) & 0xFF == 0xFF] &= np.uint32(0x807FFFFF)      # no inf / NaN inputs
        p0 = bits.view(np.float32).copy()
        bits2 = rng.integers(0, 2 ** 32, shape, dtype=np.uint64).astype(np.uint32)
        bits2[(bits2 >> 23) & 0xFF == 0xFF] &= np.uint32(0x807FFFFF)
        run(p0, bits2.view(np.float32).copy(), "random bit patterns")


def test_cpp_bench_dump_matches_python_binding(fs, tmp_path):
    """fs_bench (C++ over the C ABI) dumps raw fields + a JSON sidecar; the Python binding run with
    the same parameters must read back identical fields (same library, graph and eager paths)."""
    import os
    import subprocess
    from fluidsolver_amd import dump
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fluidsolver_amd", "bin", "fs_bench")
    prefix = str(tmp_path / "b24")
    # 2 warm-up + 3 timed (graph) + 3 profiled (eager) steps
    subprocess.run([exe, "24", "8", "3", "2", prefix], check=True, stdout=subprocess.PIPE, timeout=120)
    meta, fields = dump.read_dump(prefix)
    assert meta["frame"] == 8 and meta["nx"] == 24 and meta["jacobi_iters"] == 8
    s = fs.Solver(fs.default_params(24, 24, 24, jacobi_iters=8)).Initialize()
    s.step(8)
    for f, name in ((fs.FS_UX, "ux"), (fs.FS_UY, "uy"), (fs.FS_UZ, "uz"), (fs.FS_RHO, "rho")):
        assert np.array_equal(fields[name], s.field(f)), name
    s.close()


@pytest.mark.parametrize("brick", [0, 1])
def test_advect_displacements_around_the_brick_edge(fs, oracle, brick, monkeypatch):
    """Uniform velocities whose back-traces end 0.5 ... 3.5 cells away in each direction, so that
    waves sit on either side of the LDS brick's halo (fast path, general path with LDS taps,
    general path with global taps) on grids spanning several bricks."""
    monkeypatch.setenv("FS_ADVECT_BRICK", str(brick))
    nx, ny, nz = 70, 20, 11
    for axis in range(3):
        for speed in (2.5, -2.5, 7.5, -7.5, 12.5, -12.5, 17.5, -17.5):     # dt = 0.2: 0.5 ... 3.5 cells
            g, o = make_pair(fs, oracle, nx, ny, nz)
            st = random_fields(nx, ny, nz, seed=200 + axis)
            for a in range(3):
                st[a][...] = np.float32(speed if a == axis else 0.3)
            load_state(g, o, st)
            g.stage("advect"); o.stage("advect")
            assert_all_equal(g, o, fields=range(5), what="brick %d axis %d speed %g" % (brick, axis, speed))
            g.close(); o.close()


def test_divergence_fused_into_scalar_advection(fs, oracle):
    """Inside fs_step (one GPU) the scalar advection also writes the divergence of the advected
    velocity and the divergence stage is skipped; a stand-alone advect stage leaves div alone,
    and a divergence stage after anything that changed the velocity or div recomputes it."""
    nx, ny, nz = 40, 19, 13
    g, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=4, source_y=6.0, source_radius=4.0)
    st = random_fields(nx, ny, nz, seed=41, vel_scale=2.0)
    load_state(g, o, st)
    g.step(2); o.step(2)
    assert_all_equal(g, o, what="steps with fused divergence")
    g.stage("advect"); o.stage("advect")
    assert_all_equal(g, o, what="stand-alone advect")         # div untouched on both sides
    g.upload(fs.FS_DIV, st[6])
    g.stage("divergence")
    assert np.array_equal(g.field(fs.FS_DIV), oracle.divergence(o.get(0), o.get(1), o.get(2)))
    # a divergence stage marks div as current; a (replayed) step must clear that mark again
    g.step(); o.step()
    g.step(); o.step()
    g.stage("divergence")
    assert np.array_equal(g.field(fs.FS_DIV), oracle.divergence(o.get(0), o.get(1), o.get(2)))
    g.stage("force"); o.stage("force")
    g.stage("divergence")
    assert np.array_equal(g.field(fs.FS_DIV), oracle.divergence(o.get(0), o.get(1), o.get(2)))
    g.close(); o.close()


def _mg_pair(fs, oracle, dims, **kw):
    nx, ny, nz = dims
    g = fs.Solver(fs.default_params(nx, ny, nz, pressure_solver=fs.FS_PRESSURE_MGPCG, **kw)).Initialize()
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, pressure_solver=oracle.PRESSURE_MGPCG, **kw))
    return g, o


@pytest.mark.parametrize("smoother", ["default", "blocked", "plain", "lds2"])
@pytest.mark.parametrize("dims", [(16, 24, 16), (32, 8, 24), (64, 128, 64), (256, 192, 32)])
def test_mgpcg_steps_match_oracle(fs, oracle, dims, smoother, monkeypatch):
    """FS_PRESSURE_MGPCG (the shipped branch of Solver::Project) against the oracle's literal
    restatement: same iteration counts, same fields.  Every kernel evaluates the reference
    expressions; the dot products are binary64 sums of binary32 products on both sides (in a
    different but fixed order), narrowed to float, so results agree bit for bit unless a sum
    lands within ~1e-16 of a float rounding boundary -- asserted to 1e-5, exactness reported.
    The red/black smoother runs as plain launches per colour, four colours per pass through the
    temporally blocked kernel (by plane size; FS_MG_TB_MIN forces either), or every colour of a
    level of up to 32^3 cells in one workgroup (two kernels: FS_MG_LDS_PAD)."""
    if smoother == "lds2":
        # the older single-workgroup smoother (z and right-hand side in LDS, levels up to 8192 cells)
        monkeypatch.setenv("FS_MG_LDS_PAD", "0")
    elif smoother != "default":
        monkeypatch.setenv("FS_MG_TB_MIN", "0" if smoother == "blocked" else "1000000000")
        # "plain" also takes the small levels through one launch per colour instead of the
        # single-workgroup LDS smoother
        monkeypatch.setenv("FS_MG_NO_LDS", "1" if smoother == "plain" else "0")

    steps = 2 if dims[0] == 256 else 3 if dims[0] == 64 else 5
    g, o = _mg_pair(fs, oracle, dims, jacobi_iters=30, source_y=min(10.0, dims[1] / 3.0),
                    source_radius=min(7.0, min(dims) / 4.0))
    exact = True
    for s in range(steps):
        g.step(); o.step()
        gi, grtr = g.cg_info()
        oi, ortr = o.cg_info()
        assert gi == oi, "step %d: %d vs %d CG iterations" % (s, gi, oi)
        assert abs(grtr - ortr) <= 1e-4 * max(ortr, 1e-12)
        for f in range(7):
            a, b = g.field(f), o.get(f)
            assert rel_linf(a, b) <= 1e-5, "step %d field %d" % (s, f)
            exact &= bool(np.array_equal(a, b))
    if s == 0:
        assert not g.field(fs.FS_P).any()
    print("mgpcg %s bit-exact: %s" % (dims, exact))
    g.close(); o.close()


def test_mgpcg_vcycle_and_operator_kernels(fs, oracle):
    """One project stage from a random velocity field: exercises cg_init, A p, the V-cycle
    (smoother, restriction, prolongation on 4 levels), the vector updates and the first-solve
    pointer quirk (pressure untouched), then a second solve (pressure = solution)."""
    dims = (24, 16, 32)
    g, o = _mg_pair(fs, oracle, dims, jacobi_iters=30)
    st = random_fields(*dims, seed=5, vel_scale=1.0)
    load_state(g, o, st)
    g.stage("project"); o.stage("project")
    assert g.cg_info()[0] == o.cg_info()[0] >= 2
    assert np.array_equal(g.field(fs.FS_P), st[5])         # first solve never reaches the pressure
    assert_all_equal(g, o, fields=(5, 6), what="first solve")
    g.stage("project"); o.stage("project")
    a, b = g.field(fs.FS_P), o.get(5)
    assert rel_linf(a, b) <= 1e-5 and a.any()
    g.close(); o.close()


def test_mgpcg_rejects_unsuitable_setups(fs):
    import ctypes as C
    L = fs.load_library()
    h = C.c_void_p()
    p = fs.default_params(20, 16, 16, pressure_solver=fs.FS_PRESSURE_MGPCG)
    assert L.fs_create(C.byref(p), C.byref(h)) == -1 and b"multiple of 8" in L.fs_last_error()


@pytest.mark.parametrize("mode", ["jacobi", "shipped"])
def test_headless_program_through_the_solver_class(fs, mode):
    """fs_headless drives the reference-compatible C++ `Solver` class (Initialize / Update /
    GetDensityField on the compiled-in 64x128x64 grid); its printed sums must equal the Python
    binding's for the same configuration, also for the reference's shipped one (MGPCG +
    vorticity confinement)."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fluidsolver_amd", "bin", "fs_headless")
    frames = 4
    # no argument = the class defaults = the reference's compiled-in configuration
    out = subprocess.run([exe, str(frames)] + (["jacobi"] if mode == "jacobi" else []), check=True,
                         stdout=subprocess.PIPE, timeout=120, text=True).stdout
    assert out.count("frame: ") == frames and "frame: 0" in out and "frame: 3" in out
    m = re.search(r"sum\(rho\) ([-0-9.e+]+)\s+max\(rho\) ([-0-9.e+]+)\s+max\(uy\) ([-0-9.e+]+)", out)
    assert m, out
    kw = {}
    if mode == "shipped":
        kw = dict(flags=fs.FS_FLAG_VORT_CONFINE, pressure_solver=fs.FS_PRESSURE_MGPCG)
    s = fs.Solver(fs.default_params(64, 128, 64, **kw)).Initialize()
    s.step(frames)
    rho, uy = s.field(fs.FS_RHO), s.field(fs.FS_UY)
    assert float(m.group(1)) == pytest.approx(float(rho.sum(dtype=np.float64)), rel=1e-6)
    assert float(m.group(2)) == pytest.approx(float(rho.max()), rel=1e-6)
    assert float(m.group(3)) == pytest.approx(float(uy.max()), rel=1e-6)
    s.close()


def test_asynchronous_read_back_is_a_snapshot(fs, oracle):
    """fs_field_host_begin/_end (what replaces the blocking copy of Solver::GetDensityField,
    Solver.cu:1368-1372): the returned field is the one at the time of _begin even though more
    steps were enqueued before _end, and it is the oracle's density at that step."""
    s, o = make_pair(fs, oracle, 48, 40, 32, jacobi_iters=8)
    s.step(3)
    o.step(3)
    want = s.field(fs.FS_RHO)
    s.field_begin(fs.FS_RHO)
    s.step(4)
    got = s.field_end(fs.FS_RHO)
    assert np.array_equal(got, want)
    assert np.array_equal(got, o.get(oracle.RHO)) and got.any()
    o.close()
    assert not np.array_equal(s.field(fs.FS_RHO), want)
    with pytest.raises(fs.FsError):
        s.field_end(fs.FS_RHO)
    s.field_begin(fs.FS_UZ)
    assert np.array_equal(s.field_end(fs.FS_UZ), s.field(fs.FS_UZ))
    s.close()


@pytest.mark.parametrize("field", [3, 1, 2])   # rho, uy (ny + 1 rows), uz (nz + 1 planes)
def test_device_pointer_read_from_a_second_stream(fs, oracle, field):
    """fs_field_device (the zero-copy hand-off that replaces the host round trip of
    Solver::GetDensityField, Solver.cu:1368-1372, for a device-side consumer): after fs_sync a
    consumer reads the field through the raw pointer and the layout on a stream of its own, and
    finds the oracle's values."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    nx, ny, nz = 40, 36, 28
    s, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=8)
    s.step(3)
    o.step(3)
    s.sync()
    ptr, lay = s.device_field(field)
    assert (lay.dx, lay.dy, lay.dz) == (nx, ny + (field == 1), nz + (field == 2)) and lay.z0 == 0
    rows = lay.plane // lay.pitch
    n = lay.nz_local * lay.plane
    host = np.full(n, np.nan, np.float32)
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    assert hip.hipMemcpyAsync(host.ctypes.data, ptr, n * 4, 2, stream) == 0     # 2 = device to host
    assert hip.hipStreamSynchronize(stream) == 0
    assert hip.hipStreamDestroy(stream) == 0
    got = host.reshape(lay.nz_local, rows, lay.pitch)[:, :lay.dy, :lay.dx]
    want = o.get(field)
    assert got.shape == want.shape and np.array_equal(got, want) and want.any()
    # everything outside the field's own extent is zero (what the zero-outside gathers rely on)
    pad = host.reshape(lay.nz_local, rows, lay.pitch).copy()
    pad[:, :lay.dy, :lay.dx] = 0
    assert not pad.any()
    s.close(); o.close()


def test_pressure_solve_vcycle_and_fill_entry_points(fs, oracle):
    """The entry points behind the rest of the reference's public class (Solver.h:28-42):
    fs_stage_pressure_solve = the solve of Project() without the divergence (Conjugate / the Jacobi
    loop), fs_mg_vcycle = MG_Preconditioner on the private arrays (state-checked, no visible effect
    on the following solve), fs_field_fill = ResetGridValue's kernels."""
    nx, ny, nz = 32, 24, 16
    for solver in (fs.FS_PRESSURE_JACOBI, fs.FS_PRESSURE_MGPCG):
        g, o = make_pair(fs, oracle, nx, ny, nz, jacobi_iters=12, pressure_solver=solver)
        st = random_fields(nx, ny, nz, seed=77, vel_scale=1.0)
        load_state(g, o, st)
        for _ in range(2):       # (the first MGPCG solve does not reach the pressure field, Solver.cu:1295)
            g.stage("divergence")
            if solver == fs.FS_PRESSURE_MGPCG:
                g._check(g._L.fs_mg_vcycle(g._handle()))
            g.stage("pressure_solve")
            o.stage("project")
        if solver == fs.FS_PRESSURE_JACOBI:
            assert g._L.fs_mg_vcycle(g._handle()) == -6
            assert np.array_equal(g.field(fs.FS_P), o.get(oracle.P))
        else:
            assert g.cg_info()[0] == o.cg_info()[0]
            assert rel_linf(g.field(fs.FS_P), o.get(oracle.P)) <= 1e-5
        g.fill(fs.FS_RHO, 0.0)
        g.fill(fs.FS_UX, 0.25)
        assert not g.field(fs.FS_RHO).any()
        assert np.array_equal(g.field(fs.FS_UX), np.full((nz, ny, nx + 1), 0.25, np.float32))
        # the rows and planes beyond the field stay zero: a step from the filled state matches the oracle
        o.set(oracle.RHO, np.zeros((nz, ny, nx), np.float32))
        o.set(oracle.UX, np.full((nz, ny, nx + 1), 0.25, np.float32))
        g.step(); o.step()
        for f in range(5):
            if solver == fs.FS_PRESSURE_JACOBI:
                assert np.array_equal(g.field(f), o.get(f)), f
            else:
                assert rel_linf(g.field(f), o.get(f)) <= 1e-5, f
        g.close(); o.close()


def test_headless_program_calls_the_whole_public_class(fs):
    """fs_headless ... surface: every public method the reference class declares (Solver.h:16-44)
    is callable on the C++ `Solver`; ResetGridValue leaves density and velocity zero."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fluidsolver_amd", "bin", "fs_headless")
    for mode in ("shipped", "jacobi"):
        out = subprocess.run([exe, "3", mode, "surface"], check=True, stdout=subprocess.PIPE, timeout=120, text=True).stdout
        m = re.search(r"surface: max\(rho\) before reset ([-0-9.e+]+), nonzero after reset (\d)", out)
        assert m and float(m.group(1)) > 0 and m.group(2) == "0", out


@pytest.fixture
def fused_oracle(oracle):
    """The oracle with its contraction switch on (fso_set_fmad), switched off again afterwards."""
    oracle.set_fmad(True)
    yield oracle
    oracle.set_fmad(False)


@pytest.mark.parametrize("dims", [(21, 13, 17), (32, 24, 16), (65, 33, 9)])
def test_fused_multiply_add_mode_stages(fs, fused_oracle, dims):
    """FS_FLAG_FMAD: the buoyancy and the advection (RK4 and MacCormack, LDS bricks and the global
    path of rough fields, zero-outside walls) with lerp's left product, the RK position updates and
    the buoyancy update fused, bit for bit against the oracle's contracted mode -- and different from
    the default mode, so the flag is not a no-op."""
    oracle = fused_oracle
    nx, ny, nz = dims
    for scheme in (0, 1):
        for kind in ("smooth", "rough"):
            g, o = make_pair(fs, oracle, nx, ny, nz, scheme=scheme, flags=fs.FS_FLAG_FMAD)
            plain = fs.Solver(fs.default_params(nx, ny, nz, scheme=scheme)).Initialize()
            st = random_fields(nx, ny, nz, seed=17 + scheme, vel_scale=6.0)
            if kind == "smooth":
                st[0], st[1], st[2] = smooth_velocity(nx, ny, nz)
            load_state(g, o, st)
            for f, a in st.items():
                plain.upload(f, a)
            for stage in ("force", "advect"):
                g.stage(stage); o.stage(stage); plain.stage(stage)
                assert_all_equal(g, o, fields=range(5), what="fused %s %s scheme %d" % (stage, kind, scheme))
            assert not np.array_equal(g.field(fs.FS_RHO), plain.field(fs.FS_RHO))
            g.close(); o.close(); plain.close()


@pytest.mark.parametrize("shipped", [False, True])
def test_fused_multiply_add_mode_steps(fs, fused_oracle, shipped):
    """Whole steps in the contracted arithmetic: the Jacobi configuration bit for bit, the shipped
    one (MGPCG + vorticity confinement, whose gathers use the same lerp) to the MGPCG bar."""
    oracle = fused_oracle
    nx, ny, nz = (32, 32, 24) if shipped else (36, 28, 20)
    kw = dict(jacobi_iters=12, source_y=8.0, source_radius=5.0)
    okw = dict(kw)
    flags = fs.FS_FLAG_FMAD
    if shipped:
        kw.update(pressure_solver=fs.FS_PRESSURE_MGPCG)
        okw.update(pressure_solver=oracle.PRESSURE_MGPCG, vort_confine=1)
        flags |= fs.FS_FLAG_VORT_CONFINE
    g = fs.Solver(fs.default_params(nx, ny, nz, flags=flags, **kw)).Initialize()
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, **okw))
    st = random_fields(nx, ny, nz, seed=5, vel_scale=1.0)
    load_state(g, o, st)
    for step in range(6):
        g.step(); o.step()
        for f in range(6):
            if shipped:
                assert rel_linf(g.field(f), o.get(f)) <= 1e-5, (step, f)
            else:
                assert np.array_equal(g.field(f), o.get(f)), (step, f)
    g.close(); o.close()
