"""Known-answer tests that pin the CPU oracle (oracle/fs_oracle.c).

The reference ships no golden vectors for this path, so the oracle is pinned by cases whose
answer follows analytically from the reference expressions (file:line cited per test, relative
to CuSolver/Source/Solver.cu) and by the committed fixtures in tests/golden/.
"""
import numpy as np
import pytest

F = np.float32


def affine(shape, c0, cx, cy, cz):
    z, y, x = np.meshgrid(*(np.arange(n) for n in shape), indexing="ij")
    return (c0 + cx * x + cy * y + cz * z).astype(F)


def test_trilerp_affine_field_interior_exact(oracle):
    """trilerp (:52-113) of an affine field with dyadic coefficients is exact in the interior."""
    f = affine((9, 10, 11), 1.0, 0.5, 0.25, 2.0)
    for pos in [(3.25, 4.5, 2.75), (5.0, 5.0, 5.0), (1.125, 7.875, 6.5)]:
        got = oracle.trilerp(f, [F(v) for v in pos], (F(0), F(0), F(0)), clamp=False)
        want = 1.0 + 0.5 * pos[0] + 0.25 * pos[1] + 2.0 * pos[2]
        assert got == F(want)


def test_trilerp_zero_outside_vs_clamp(oracle):
    """cg_sample (:37-42) returns 0 outside, sample (:30-34) replicates the edge."""
    f = np.full((4, 4, 4), 3.0, F)
    pos = [F(-0.5), F(1.0), F(1.0)]
    assert oracle.trilerp(f, pos, (F(0), F(0), F(0)), clamp=True) == F(3.0)
    # x = -0.5: ix = -1 (outside -> 0), fx = 0.5  =>  0*0.5 + 3*0.5
    assert oracle.trilerp(f, pos, (F(0), F(0), F(0)), clamp=False) == F(1.5)
    assert oracle.trilerp(f, [F(10.0), F(1.0), F(1.0)], (F(0), F(0), F(0)), clamp=False) == F(0.0)


def test_trilerp_quirk_window_and_negative_integers(oracle):
    """:58-82: for x in (-1e-6, 0) the fix-up does not fire (ix = 0, fx = x + 1 ~ 1), so the value
    is ~field[1] instead of ~field[0]; for x = -2 exactly, ix = -3 and fx = 1."""
    f = affine((3, 3, 8), 10.0, 1.0, 0.0, 0.0)     # f[x] = 10 + x
    y = z = F(1.0)
    x = F(-5e-7)
    fx = F(x - F(np.floor(x)))                      # = x + 1 rounded to binary32
    want = F(F(F(10.0) * F(F(1) - fx)) + F(F(11.0) * fx))
    got = oracle.trilerp(f, [x, y, z], (F(0), F(0), F(0)), clamp=False)
    assert got == want and abs(float(got) - 11.0) < 1e-5
    # just below the window the fix-up fires: ix = -1 (outside, 0), fx = x + 1
    x2 = F(-2e-6)
    fx2 = F(x2 - F(-1.0))
    want2 = F(F(F(0.0) * F(F(1) - fx2)) + F(F(10.0) * fx2))
    assert oracle.trilerp(f, [x2, y, z], (F(0), F(0), F(0)), clamp=False) == want2
    # exact negative integer: both taps outside -> 0 (ix = -3, ix+1 = -2)
    assert oracle.trilerp(f, [F(-2.0), y, z], (F(0), F(0), F(0)), clamp=False) == F(0.0)
    # x = -1 exactly: ix = -2, fx = 1: taps -2 and -1, both outside -> 0 (not f[0]!)
    assert oracle.trilerp(f, [F(-1.0), y, z], (F(0), F(0), F(0)), clamp=False) == F(0.0)


def test_float_threshold_equals_double_threshold():
    """The HIP kernels test `x < -1e-6f` in binary32 where the reference compares against the
    double literal (:66); the two agree for every float because -1e-6f is the float neighbour of
    -1e-6 on the side of zero."""
    lit = F(-1e-6)
    assert float(lit) > -1e-6
    below = np.nextafter(lit, F(-1))
    assert float(below) < -1e-6
    for x in (lit, below, np.nextafter(below, F(-1)), np.nextafter(lit, F(0))):
        assert (float(x) < -1e-6) == bool(x < lit)


def test_zero_velocity_advection_is_identity(oracle):
    """:139-169, :370-403: u = 0 => the back-trace returns the sample point and the gather at a
    sample point returns the stored value bit-for-bit."""
    rng = np.random.default_rng(1)
    nx, ny, nz = 9, 7, 6
    ux = np.zeros((nz, ny, nx + 1), F); uy = np.zeros((nz, ny + 1, nx), F); uz = np.zeros((nz + 1, ny, nx), F)
    rho = rng.random((nz, ny, nx), dtype=F)
    for scheme in (0, 1):
        assert np.array_equal(oracle.semilag(rho, ux, uy, uz, F(0.2), 0, scheme), rho)
    c = oracle.backtrace(ux, uy, uz, [F(2.5), F(3.5), F(1.5)], F(0.2))
    assert list(c) == [F(2.5), F(3.5), F(1.5)]


def test_uniform_velocity_integer_shift(oracle):
    """Uniform u with dt*u = 1 cell: RK4 and MacCormack trace back exactly one cell, so interior
    values shift by one cell (outside is 0)."""
    nx, ny, nz = 12, 6, 6
    rng = np.random.default_rng(2)
    rho = rng.random((nz, ny, nx), dtype=F)
    ux = np.full((nz, ny, nx + 1), 4.0, F); uy = np.zeros((nz, ny + 1, nx), F); uz = np.zeros((nz + 1, ny, nx), F)
    for scheme in (0, 1):
        out = oracle.semilag(rho, ux, uy, uz, F(0.25), 0, scheme)
        # far enough from the walls the trace only sees u = 4
        assert np.array_equal(out[1:-1, 1:-1, 3:-3], rho[1:-1, 1:-1, 2:-4])


def test_divergence_of_linear_velocity(oracle):
    """:406-423: ux = i, uy = uz = 0 => div = 1 everywhere (h = 1, no scaling)."""
    nx, ny, nz = 8, 5, 4
    ux = affine((nz, ny, nx + 1), 0.0, 1.0, 0.0, 0.0)
    uy = np.zeros((nz, ny + 1, nx), F); uz = np.zeros((nz + 1, ny, nx), F)
    assert np.array_equal(oracle.divergence(ux, uy, uz), np.ones((nz, ny, nx), F))
    # the summation order is ur + uf + ut - ul - ubh - ubo
    rng = np.random.default_rng(3)
    ux = rng.standard_normal(ux.shape).astype(F); uy = rng.standard_normal(uy.shape).astype(F)
    uz = rng.standard_normal(uz.shape).astype(F)
    want = ((((ux[:, :, 1:] + uz[1:]) + uy[:, 1:, :]) - ux[:, :, :-1]) - uz[:-1]) - uy[:, :-1, :]
    assert np.array_equal(oracle.divergence(ux, uy, uz), want)


def test_jacobi_constant_pressure_fixed_point(oracle):
    """:426-443: constant p, zero div is a fixed point only because out-of-range neighbours are
    clamped (Neumann walls); the residual of it is 0."""
    p = np.full((5, 6, 7), 2.5, F)
    div = np.zeros_like(p)
    newp, r = oracle.jacobi(p, div)
    assert np.array_equal(newp, p)
    assert oracle.sum_sq(r) == 0.0


def test_jacobi_expression_order(oracle):
    """:441 (pl + pr + pbh + pf + pbo + pt - div) / 6 and :442 residual, on random data."""
    rng = np.random.default_rng(4)
    p = rng.standard_normal((6, 5, 7)).astype(F)
    div = rng.standard_normal(p.shape).astype(F)
    pad = np.pad(p, 1, mode="edge")
    pl, pr = pad[1:-1, 1:-1, :-2], pad[1:-1, 1:-1, 2:]
    pbo, pt = pad[1:-1, :-2, 1:-1], pad[1:-1, 2:, 1:-1]
    pbh, pf = pad[:-2, 1:-1, 1:-1], pad[2:, 1:-1, 1:-1]
    want = ((((((pl + pr) + pbh) + pf) + pbo) + pt) - div) / F(6)
    rwant = (((((((div + F(6) * p) - pl) - pr) - pbh) - pf) - pbo) - pt)
    newp, r = oracle.jacobi(p, div)
    assert np.array_equal(newp, want.astype(F))
    assert np.array_equal(r, rwant.astype(F))
    assert oracle.sum_sq(r) == pytest.approx(float(np.sum(r.astype(np.float64) ** 2)), rel=1e-12)


def test_gradient_of_linear_pressure(oracle):
    """:446-483: p = i => interior ux faces lose exactly 1, the wall faces i = 0 and i = nx see two
    equal clamped samples and stay unchanged; uy, uz unchanged."""
    nx, ny, nz = 7, 4, 5
    p = affine((nz, ny, nx), 0.0, 1.0, 0.0, 0.0)
    ux = np.full((nz, ny, nx + 1), 10.0, F); uy = np.full((nz, ny + 1, nx), 10.0, F)
    uz = np.full((nz + 1, ny, nx), 10.0, F)
    oracle.gradient(ux, uy, uz, p)
    assert np.all(ux[:, :, 1:-1] == F(9.0)) and np.all(ux[:, :, 0] == F(10.0)) and np.all(ux[:, :, -1] == F(10.0))
    assert np.all(uy == F(10.0)) and np.all(uz == F(10.0))


def test_source_overwrites_inside_sphere(oracle):
    """:295-307: cells with (i-sx)^2+(j-sy)^2+(k-sz)^2 <= R^2 (integer cell indices) are SET to
    rho0 / T0, others untouched."""
    n = 20
    rho = np.full((n, n, n), 0.5, F); T = np.full((n, n, n), -1.0, F)
    oracle.source(rho, T, F(10), F(8), F(9), F(3), F(0.1), F(2.0))
    z, y, x = np.meshgrid(*(np.arange(n),) * 3, indexing="ij")
    inside = (x - 10) ** 2 + (y - 8) ** 2 + (z - 9) ** 2 <= 9
    assert np.all(rho[inside] == F(0.1)) and np.all(T[inside] == F(2.0))
    assert np.all(rho[~inside] == F(0.5)) and np.all(T[~inside] == F(-1.0))
    assert inside.sum() == 123     # lattice points in a ball of radius 3


def test_force_is_mean_of_vertical_neighbours(oracle):
    """:346-368: uy(i,j,k) += (0.5*T(j-1) + 0.5*T(j) - T_env) * dt with j clamped at both walls."""
    rng = np.random.default_rng(5)
    nx, ny, nz = 5, 6, 4
    T = rng.standard_normal((nz, ny, nx)).astype(F)
    uy = np.zeros((nz, ny + 1, nx), F)
    oracle.force_uy(uy, T, F(0.2), F(0.25))
    Tp = np.concatenate([T[:, :1], T, T[:, -1:]], axis=1)     # clamp j-1 and j
    that = Tp[:, :-1] * F(0.5) + Tp[:, 1:] * F(0.5)
    want = (that - F(0.25)) * F(0.2)
    assert np.array_equal(uy, want.astype(F))


def test_step_order_scalars_use_new_velocity(oracle):
    """:1145-1165: T and rho are advected with the already advected velocity."""
    n = 12
    p = oracle.default_params(n, n, n, jacobi_iters=0, source_y=4.0, source_radius=3.0)
    a = oracle.Oracle(p)
    a.step(3)
    b = oracle.Oracle(p)
    b.step(2)
    b.stage("source"); b.stage("force")
    ux, uy, uz = b.get(0), b.get(1), b.get(2)
    nux = oracle.semilag(ux, ux, uy, uz, p.dt, 1)
    nuy = oracle.semilag(uy, ux, uy, uz, p.dt, 2)
    nuz = oracle.semilag(uz, ux, uy, uz, p.dt, 3)
    rho_new = oracle.semilag(b.get(3), nux, nuy, nuz, p.dt, 0)
    assert np.array_equal(a.get(3), rho_new)     # jacobi_iters = 0: p stays 0, gradient is a no-op
    rho_old_vel = oracle.semilag(b.get(3), ux, uy, uz, p.dt, 0)
    assert not np.array_equal(rho_new, rho_old_vel)


def test_pressure_warm_start_and_frame_counter(oracle):
    """:927-928, :1180-1184: p is zeroed only at init and carried from step to step."""
    n = 12
    p = oracle.default_params(n, n, n, jacobi_iters=5, source_y=4.0, source_radius=3.0)
    o = oracle.Oracle(p)
    o.step(2)
    p2 = o.get(5)
    assert float(np.abs(p2).max()) > 0
    o.stage("source"); o.stage("force"); o.stage("advect")
    d = oracle.divergence(o.get(0), o.get(1), o.get(2))
    q = p2
    for _ in range(5):
        q, _ = oracle.jacobi(q, d, want_r=False)
    o.stage("project")
    assert np.array_equal(o.get(5), q)


def test_default_params_match_reference_header(oracle):
    """CuSolver/Include/Solver.h:58-81."""
    p = oracle.default_params(64, 128, 64)
    assert (p.nx, p.ny, p.nz) == (64, 128, 64)
    assert p.dt == F(0.2) and p.jacobi_iters == 30 and p.scheme == 1
    assert (p.source_x, p.source_y, p.source_z, p.source_radius) == (32.0, 10.0, 32.0, 7.0)
    assert p.rho0 == F(0.1) and p.temperature0 == 2.0 and p.temperature_env == 0.0


# ------------------------------------------------------------ vorticity confinement variant

def test_vorticity_of_rigid_rotation_and_layout(oracle):
    """FindVortx/y/z (Solver.cu:485-525): for u = (-y', x', 0) sampled on the staggered grid the
    z component is  ut - ubo - ur + ul = -1 - 1 = -2  in the interior, the others 0; the values sit
    at the cell index i + j*nx + k*nx*ny and the rest of the (nx+1)(ny+1)(nz+1) array stays 0."""
    nx, ny, nz = 6, 5, 4
    ux = np.zeros((nz, ny, nx + 1), np.float32); uy = np.zeros((nz, ny + 1, nx), np.float32)
    uz = np.zeros((nz + 1, ny, nx), np.float32)
    ux[:, :, :] = -(np.arange(ny, dtype=np.float32) + 0.5)[None, :, None]     # -y at (i, j+.5)
    uy[:, :, :] = (np.arange(nx, dtype=np.float32) + 0.5)[None, None, :]      # +x at (i+.5, j)
    vx, vy, vz = oracle.find_vort(ux, uy, uz)
    n = nx * ny * nz
    assert vx.size == (nx + 1) * (ny + 1) * (nz + 1)
    assert not vx[n:].any() and not vy[n:].any() and not vz[n:].any()
    assert not vx[:n].any() and not vy[:n].any()
    z = vz[:n].reshape(nz, ny, nx)
    assert np.all(z[:, 1:, 1:] == -2.0)
    # clamped neighbours at i = 0 / j = 0 (Solver.cu:520, 522): one of the two differences vanishes
    assert np.all(z[:, 1:, 0] == -1.0) and np.all(z[:, 0, 1:] == -1.0) and np.all(z[:, 0, 0] == 0.0)


def test_confinement_guard_and_zero_vorticity(oracle):
    """ApplyVortConfinement (Solver.cu:540-605): zero vorticity gives n = sqrt(1e-5) and a zero
    force; with a non-zero field only faces with 2 < index < launch_dim - 2 change (:548)."""
    nx, ny, nz = 9, 8, 10
    rng = np.random.default_rng(5)
    for d, shape in ((1, (nz, ny, nx + 1)), (2, (nz, ny + 1, nx)), (3, (nz + 1, ny, nx))):
        u0 = rng.standard_normal(shape).astype(np.float32)
        zero = np.zeros(oracle.vort_size(nx, ny, nz), np.float32)
        u = u0.copy()
        oracle.vort_confine(u, zero, zero, zero, (nx, ny, nz), d, 0.1)
        assert np.array_equal(u, u0)
        vx, vy, vz = (rng.standard_normal(zero.size).astype(np.float32) for _ in range(3))
        u = u0.copy()
        oracle.vort_confine(u, vx, vy, vz, (nx, ny, nz), d, 0.1)
        changed = u != u0
        dz, dy, dx = shape
        inner = np.zeros(shape, bool)
        inner[3:dz - 2, 3:dy - 2, 3:dx - 2] = True
        assert not changed[~inner].any()
        assert changed[inner].mean() > 0.9


def test_confinement_step_order(oracle):
    """Solver.cu:1031-1082: vorticity from the post-buoyancy velocity, advection, confinement on
    the advected velocity with that vorticity, then the projection."""
    kw = dict(jacobi_iters=5, source_y=5.0, source_radius=3.0, vort_confine=1)
    a = oracle.Oracle(oracle.default_params(12, 14, 10, **kw))
    b = oracle.Oracle(oracle.default_params(12, 14, 10, **kw))
    for _ in range(3):
        a.step()
        b.stage("source"); b.stage("force"); b.stage("vorticity"); b.stage("advect"); b.stage("confine")
        b.stage("project"); b.stage("gradient")
    for f in range(6):
        assert np.array_equal(a.get(f), b.get(f))
    a.close(); b.close()


# ------------------------------------------------------------ MGPCG

def test_apply_a_is_the_dirichlet_laplacian(oracle):
    """ComputeAp (Solver.cu:789-799): 6p - neighbours with zero outside: constant p = 1 gives 0 in
    the interior, 1 on faces, 2 on edges, 3 in corners."""
    p = np.ones((6, 5, 7), np.float32)
    a = oracle.apply_a(p)
    assert np.all(a[1:-1, 1:-1, 1:-1] == 0) and a[0, 0, 0] == 3 and a[0, 0, 3] == 2 and a[0, 2, 3] == 1


def test_multigrid_levels_and_transfer_operators(oracle):
    """mg_space (Solver.cu:1120-1125); Restrict adds half of each of the 8 children's residuals
    (with z = 0 the residual is r); Prolongate injects the parent into its children."""
    import ctypes as C
    assert oracle.mg_space(64, 128, 64, 4) == 524288 + 65536 + 8192 + 1024
    nx, ny, nz = 8, 8, 8
    n = nx * ny * nz
    r = np.zeros(oracle.mg_space(nx, ny, nz, 2), np.float32)
    z = np.zeros_like(r)
    r[:n] = 1.0
    L = oracle.lib()
    L.fso_mg_restrict(oracle._p(r), oracle._p(z), 0, nx, ny, nz)
    assert np.all(r[n:] == 4.0)                      # 8 children x 0.5
    z[n:] = np.arange(n // 8, dtype=np.float32)
    L.fso_mg_prolongate(oracle._p(z), 0, nx, ny, nz)
    fine = z[:n].reshape(nz, ny, nx)
    coarse = z[n:].reshape(nz // 2, ny // 2, nx // 2)
    assert np.array_equal(fine, coarse.repeat(2, 0).repeat(2, 1).repeat(2, 2))


def test_red_black_smoother_fixed_point(oracle):
    """Smooth (Solver.cu:848-860): for r = A z* the exact z* is a fixed point of both colours."""
    rng = np.random.default_rng(9)
    zs = rng.integers(-8, 9, (8, 8, 8)).astype(np.float32)        # small integers: exact arithmetic / 6
    zs *= 6.0
    r = oracle.apply_a(zs)
    z = zs.copy().reshape(-1)
    rr = r.reshape(-1).copy()
    L = oracle.lib()
    for phase in (0, 1):
        L.fso_mg_smooth(oracle._p(rr), oracle._p(z), 0, phase, 8, 8, 8)
    assert np.array_equal(z.reshape(8, 8, 8), zs)


def test_mgpcg_solves_the_system_and_keeps_the_pointer_quirk(oracle):
    """Conjugate (Solver.cu:1226-1297): r.r falls below iter_precision within a few iterations and
    A p = -div; the first solve does not reach the pressure field (swap of a by-value pointer at
    :1295), later ones do."""
    kw = dict(jacobi_iters=30, source_y=10.0, source_radius=4.0, pressure_solver=oracle.PRESSURE_MGPCG)
    o = oracle.Oracle(oracle.default_params(16, 24, 16, **kw))
    o.step()
    assert not o.get(5).any()                        # frame 0: unprojected
    it, rtr = o.cg_info()
    assert 1 <= it <= 10 and rtr < 1e-3
    o.step()
    p, div = o.get(5), o.get(6)
    assert p.any()
    assert float(np.abs(oracle.apply_a(p) + div).max()) < 5e-3 * float(np.abs(div).max()) + 1e-4
    o.close()
    with pytest.raises(MemoryError):                 # dims must be multiples of 2^(levels-1)
        oracle.Oracle(oracle.default_params(12, 24, 16, **kw))


def test_contraction_sensitivity_of_the_64_cube_configuration(oracle):
    """How far can fused multiply-adds move the fields?  The reference is built by nvcc, whose
    default -fmad=true contracts x*y + z wherever it likes; the oracle (and every GPU kernel)
    evaluates the uncontracted expressions.  BASELINE configuration 1 (64^3, 100 steps, 20 sweeps)
    stepped both ways: the density stays within the 1e-4 the north-star allows, the velocity
    components do not (a few 1e-3 of their maxima after 100 steps).  So agreement with a reference
    BINARY to 1e-4 on velocity would need its exact contraction pattern, which the source does not
    determine; what this repository pins is bit-equality with the uncontracted meaning."""
    def run(fmad):
        oracle.set_fmad(fmad)
        try:
            o = oracle.Oracle(oracle.default_params(64, 64, 64, jacobi_iters=20))
            o.step(100)
            out = [o.get(f).copy() for f in range(5)]
            o.close()
        finally:
            oracle.set_fmad(False)
        return out
    strict, fused = run(False), run(True)
    rel = [float(np.abs(a - b).max() / np.abs(a).max()) for a, b in zip(strict, fused)]
    ux, uy, uz, rho, T = rel
    assert 0 < rho < 2e-4 and 0 < T < 2e-4          # measured 9.4e-5
    assert max(ux, uy, uz) < 2e-2                    # measured 3.7e-3, 6.8e-4, 2.8e-3
    assert max(ux, uy, uz) > 1e-4                    # the point of the test: contraction is visible
    # and the switch is really off again: a strict run reproduces itself
    again = run(False)
    assert all(np.array_equal(a, b) for a, b in zip(strict, again))
