"""Properties that do not depend on the grid size, checked at the benchmark's full sizes.

The CPU oracle needs minutes per step at 512^3, so at that size the HIP path is checked against
itself through properties that hold for every size and that the small-grid parity tests tie to
the oracle: a temporally blocked Jacobi pass equals its single sweeps, a replayed graph equals
eager launches, both LDS brick shapes return the same advection, a sweep is exactly linear under
scaling by two, and zero velocity advects nothing.  All comparisons are bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = range(6)   # ux uy uz rho T p


def plume(fs, n, steps, **kw):
    s = fs.Solver(fs.default_params(n, n, n, **kw)).Initialize()
    s.step(steps)
    out = [s.field(f) for f in FIELDS]
    s.close()
    return out


@pytest.mark.parametrize("n,sweeps,steps", [(512, 12, 2), (256, 80, 4)])
def test_blocked_graph_path_equals_single_sweeps_launched_eagerly(fs, n, sweeps, steps):
    """Default path (4 sweeps per pass, hipGraph replay) against one launch per sweep without a
    graph, from the zero state with the source on: BASELINE sizes 512^3 and 256^3."""
    a = plume(fs, n, steps, jacobi_iters=sweeps)
    b = plume(fs, n, steps, jacobi_iters=sweeps, flags=fs.FS_FLAG_JACOBI_NAIVE | fs.FS_FLAG_NO_GRAPH)
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), "field %d" % f
    assert a[5].any() and a[0].any()


def test_both_brick_shapes_advect_alike_at_512(fs, monkeypatch):
    monkeypatch.setenv("FS_ADVECT_BRICK", "1")
    a = plume(fs, 512, 2, jacobi_iters=8, flags=fs.FS_FLAG_NO_GRAPH)
    monkeypatch.setenv("FS_ADVECT_BRICK", "0")
    b = plume(fs, 512, 2, jacobi_iters=8, flags=fs.FS_FLAG_NO_GRAPH)
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), "field %d" % f


def test_jacobi_is_exactly_linear_under_scaling_by_two_at_512(fs):
    """(p, div) -> (2p, 2div) doubles every sum and every quotient exactly (no overflow, no
    subnormals here), so eight sweeps of the doubled problem are the doubled result."""
    n = 512
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal((n, n, n), dtype=np.float32)
    d0 = rng.standard_normal((n, n, n), dtype=np.float32)
    s = fs.Solver(fs.default_params(n, n, n, jacobi_iters=8)).Initialize()
    s.upload(fs.FS_P, p0); s.upload(fs.FS_DIV, d0)
    s.stage("jacobi", 8)
    p1 = s.field(fs.FS_P)
    s.upload(fs.FS_P, p0 * 2); s.upload(fs.FS_DIV, d0 * 2)
    s.stage("jacobi", 8)
    p2 = s.field(fs.FS_P)
    s.close()
    assert np.array_equal(p2, p1 * 2)
    assert not np.array_equal(p1, p0)


def test_zero_velocity_advects_nothing_at_512(fs):
    n = 512
    rng = np.random.default_rng(4)
    rho = rng.random((n, n, n), dtype=np.float32)
    T = rng.standard_normal((n, n, n), dtype=np.float32)
    s = fs.Solver(fs.default_params(n, n, n)).Initialize()
    s.upload(fs.FS_RHO, rho); s.upload(fs.FS_T, T)
    s.stage("advect")
    assert np.array_equal(s.field(fs.FS_RHO), rho)
    assert np.array_equal(s.field(fs.FS_T), T)
    for f in (fs.FS_UX, fs.FS_UY, fs.FS_UZ):
        assert not s.field(f).any()
    s.close()


@pytest.mark.parametrize("n,steps", [(256, 3), (512, 2)])
def test_full_size_steps_match_the_oracle(fs, oracle, n, steps):
    """BASELINE's single-GPU configurations themselves (256^3 and 512^3, 80 sweeps, MacCormack)
    against the CPU oracle for the first steps of the plume: every field, bit for bit."""
    g = fs.Solver(fs.default_params(n, n, n, jacobi_iters=80)).Initialize()
    o = oracle.Oracle(oracle.default_params(n, n, n, jacobi_iters=80))
    for _ in range(steps):
        g.step()
        o.step()
    for f in range(7):
        assert np.array_equal(g.field(f), o.get(f)), "field %d" % f
    assert g.field(fs.FS_UY).any()
    g.close()
    o.close()


def test_developed_fields_at_512_match_the_oracle(fs, oracle):
    """BASELINE's largest single-GPU grid with data that is non-zero everywhere, stage by stage
    against the oracle, every value: buoyancy from a random temperature (ForceKernelUy,
    Solver.cu:346-368), the advection of all five fields in a smooth swirl whose back-traces span
    0-3 cells and leave the box at every wall (SemiLagKernel, Solver.cu:370-403), 14 Jacobi sweeps
    from random pressure and divergence (Solver.cu:426-443; one pass of six and two of four where
    the six-sweep passes are on, else three of four and two single sweeps) and the gradient subtraction of that
    pressure (Solver.cu:446-483).  An addressing error at large offsets cannot hide behind zeros."""
    import bench
    n = 512
    g = fs.Solver(fs.default_params(n, n, n, jacobi_iters=14)).Initialize()
    o = oracle.Oracle(oracle.default_params(n, n, n, jacobi_iters=14))
    st = bench.swirl_fields(n, seed=21)
    rng = np.random.default_rng(22)
    st[5] = rng.standard_normal((n, n, n), dtype=np.float32)
    st[6] = rng.standard_normal((n, n, n), dtype=np.float32)
    for f, a in st.items():
        g.upload(f, a)
        o.set(f, a)
    g.stage("force"); o.stage("force")
    assert np.array_equal(g.field(fs.FS_UY), o.get(oracle.UY)), "buoyancy"
    g.stage("advect"); o.stage("advect")
    for f in range(5):
        got, want = g.field(f), o.get(f)
        assert np.array_equal(got, want), "advection, field %d: %d values differ" % (f, int((got != want).sum()))
        assert np.count_nonzero(want) > 0.99 * want.size
    p = st[5]
    for _ in range(14):
        p, _r = oracle.jacobi(p, st[6], want_r=False)
    g.stage("jacobi", 14)
    got = g.field(fs.FS_P)
    assert np.array_equal(got, p), "14 sweeps: %d values differ" % int((got != p).sum())
    o.set(oracle.P, p)
    g.stage("gradient"); o.stage("gradient")
    for f in range(3):
        assert np.array_equal(g.field(f), o.get(f)), "gradient, component %d" % f
    g.close()
    o.close()
