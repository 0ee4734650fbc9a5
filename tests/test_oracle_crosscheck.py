"""A second, independent transcription of the advection (numpy float32 scalars, plain loops) against
the C oracle, bit for bit.

The reference holds no vectors for this path, so the oracle is pinned by literal fidelity only.
This file restates the same reference lines a second time, in another language and without
looking at the C code's structure: CuSolver/Source/Solver.cu sample/cg_sample :30-42, trilerp
:52-113, RK4 :139-161, MacCormack :163-169, SemiLagKernel :370-403, Solver::Advect :1145-1165;
CuSolver/Include/Math.cuh lerp :57-60 and the float3 operators :80-118.  Two transcriptions that
agree on random fields (negative coordinates, traces leaving the box included) leave little room
for a slip in association order, truncation or the zero-outside rule in either.
"""
import math

import numpy as np
import pytest

F = np.float32


def cg_sample(field, x, y, z):   # :37-42, field indexed [z, y, x]
    dz, dy, dx = field.shape
    if x < 0 or x >= dx or y < 0 or y >= dy or z < 0 or z >= dz:
        return F(0)
    return field[z, y, x]


def lerp(a, b, s):   # Math.cuh:57-60
    return F(F(a * F(F(1) - s)) + F(b * s))


def split(c):   # :58-82 for one axis
    i = int(c)                                   # truncation toward zero
    f = F(c - F(math.floor(c)))
    if float(c) < -1e-6:                         # float promoted to double against the literal
        i -= 1
        f = F(c - F(i))
    return i, f


def trilerp(field, pos, off):   # :52-113 with b_minmax_sampe = false
    ix, fx = split(F(pos[0] - off[0]))
    iy, fy = split(F(pos[1] - off[1]))
    iz, fz = split(F(pos[2] - off[2]))
    a = cg_sample(field, ix, iy, iz)
    b = cg_sample(field, ix + 1, iy, iz)
    c = cg_sample(field, ix, iy + 1, iz)
    d = cg_sample(field, ix + 1, iy + 1, iz)
    e = cg_sample(field, ix, iy, iz + 1)
    f = cg_sample(field, ix + 1, iy, iz + 1)
    g = cg_sample(field, ix, iy + 1, iz + 1)
    h = cg_sample(field, ix + 1, iy + 1, iz + 1)
    l1 = lerp(lerp(a, b, fx), lerp(c, d, fx), fy)
    l2 = lerp(lerp(e, f, fx), lerp(g, h, fx), fy)
    return lerp(l1, l2, fz)


OFF = {0: (F(.5), F(.5), F(.5)), 1: (F(0), F(.5), F(.5)), 2: (F(.5), F(0), F(.5)), 3: (F(.5), F(.5), F(0))}


def velocity(ux, uy, uz, pos):   # the three gathers of :142-144
    return [trilerp(ux, pos, OFF[1]), trilerp(uy, pos, OFF[2]), trilerp(uz, pos, OFF[3])]


def rk4(ux, uy, uz, pos, dt):   # :139-161
    u1 = velocity(ux, uy, uz, pos)
    h = F(F(0.5) * dt)
    p1 = [F(pos[q] - F(h * u1[q])) for q in range(3)]
    u2 = velocity(ux, uy, uz, p1)
    p2 = [F(pos[q] - F(h * u2[q])) for q in range(3)]
    u3 = velocity(ux, uy, uz, p2)
    p3 = [F(pos[q] - F(dt * u3[q])) for q in range(3)]
    u4 = velocity(ux, uy, uz, p3)
    out = []
    for q in range(3):   # pos - dt * (u1 + 2.0 * u2 + 2.0 * u3 + u4) / 6.0, float operators throughout
        s = F(F(F(u1[q] + F(F(2) * u2[q])) + F(F(2) * u3[q])) + u4[q])
        out.append(F(pos[q] - F(F(dt * s) / F(6))))
    return out


def maccormack(ux, uy, uz, pos, dt):   # :163-169
    pred = rk4(ux, uy, uz, pos, dt)
    refl = rk4(ux, uy, uz, pred, F(-dt))
    return [F(pred[q] + F(F(pos[q] - refl[q]) / F(2))) for q in range(3)]


def semilag(field, ux, uy, uz, dt, direction, mac):   # :370-403
    out = np.empty_like(field)
    dz, dy, dx = field.shape
    o = OFF[direction]
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                pos = [F(F(i) + o[0]), F(F(j) + o[1]), F(F(k) + o[2])]
                coord = maccormack(ux, uy, uz, pos, dt) if mac else rk4(ux, uy, uz, pos, dt)
                out[k, j, i] = trilerp(field, coord, o)
    return out


def advect(st, dt, mac):   # Solver::Advect :1145-1165
    ux, uy, uz = st[0], st[1], st[2]
    nux = semilag(ux, ux, uy, uz, dt, 1, mac)
    nuy = semilag(uy, ux, uy, uz, dt, 2, mac)
    nuz = semilag(uz, ux, uy, uz, dt, 3, mac)
    T = semilag(st[4], nux, nuy, nuz, dt, 0, mac)      # temperature first, with the NEW velocity
    rho = semilag(st[3], nux, nuy, nuz, dt, 0, mac)
    return {0: nux, 1: nuy, 2: nuz, 3: rho, 4: T}


@pytest.mark.parametrize("scheme", [1, 0])
@pytest.mark.parametrize("dims,scale", [((6, 5, 4), 1.5), ((4, 7, 5), 6.0)])
def test_second_transcription_of_the_advection_agrees_with_the_oracle(oracle, dims, scale, scheme):
    from conftest import random_fields
    nx, ny, nz = dims
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, scheme=scheme))
    st = random_fields(nx, ny, nz, seed=77 + nx + scheme, vel_scale=scale)
    for fid, arr in st.items():
        o.set(fid, arr)
    o.stage("advect")
    want = advect(st, F(o.params.dt), scheme == 1)
    for fid, arr in want.items():
        got = o.get(fid)
        assert got.shape == arr.shape
        assert np.array_equal(got, arr), "field %d: %d of %d differ" % (fid, int((got != arr).sum()), arr.size)
    o.close()


def sample(field, x, y, z):   # :30-34 with Math.cuh minmax :33-39
    dz, dy, dx = field.shape
    return field[max(0, min(z, dz - 1)), max(0, min(y, dy - 1)), max(0, min(x, dx - 1))]


def trilerp_clamped(field, pos, off):   # :52-113 with b_minmax_sampe = true
    ix, fx = split(F(pos[0] - off[0]))
    iy, fy = split(F(pos[1] - off[1]))
    iz, fz = split(F(pos[2] - off[2]))
    tap = lambda a, b, c: sample(field, ix + a, iy + b, iz + c)
    l1 = lerp(lerp(tap(0, 0, 0), tap(1, 0, 0), fx), lerp(tap(0, 1, 0), tap(1, 1, 0), fx), fy)
    l2 = lerp(lerp(tap(0, 0, 1), tap(1, 0, 1), fx), lerp(tap(0, 1, 1), tap(1, 1, 1), fx), fy)
    return lerp(l1, l2, fz)


def force(uy, T, dt, T_env):   # ForceKernelUy :346-368, launched over the uy faces
    out = uy.copy()
    dz, dy, dx = uy.shape
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                pos = [F(F(i) + F(.5)), F(j), F(F(k) + F(.5))]
                t = trilerp_clamped(T, pos, OFF[0])
                out[k, j, i] = F(uy[k, j, i] + F(F(t - T_env) * dt))
    return out


def divergence(ux, uy, uz):   # DivergenceKernel :406-423
    dz, dy, dx = uz.shape[0] - 1, uy.shape[1] - 1, ux.shape[2] - 1
    out = np.empty((dz, dy, dx), F)
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                ul, ur = sample(ux, i, j, k), sample(ux, i + 1, j, k)
                ubo, ut = sample(uy, i, j, k), sample(uy, i, j + 1, k)
                ubh, uf = sample(uz, i, j, k), sample(uz, i, j, k + 1)
                out[k, j, i] = F(F(F(F(F(ur + uf) + ut) - ul) - ubh) - ubo)
    return out


def jacobi(p, div):   # JacobiKernel :426-443, one sweep
    out = np.empty_like(p)
    dz, dy, dx = p.shape
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                pl, pr = sample(p, i - 1, j, k), sample(p, i + 1, j, k)
                pbo, pt = sample(p, i, j - 1, k), sample(p, i, j + 1, k)
                pbh, pf = sample(p, i, j, k - 1), sample(p, i, j, k + 1)
                s = F(F(F(F(F(F(pl + pr) + pbh) + pf) + pbo) + pt) - div[k, j, i])
                out[k, j, i] = F(s / F(6))
    return out


def gradient(ux, uy, uz, p):   # ApplyGradientUx/Uy/Uz :446-483
    ox, oy, oz = ux.copy(), uy.copy(), uz.copy()
    for arr, out, d in ((ux, ox, (1, 0, 0)), (uy, oy, (0, 1, 0)), (uz, oz, (0, 0, 1))):
        dz, dy, dx = arr.shape
        for k in range(dz):
            for j in range(dy):
                for i in range(dx):
                    lo = sample(p, i - d[0], j - d[1], k - d[2])
                    hi = sample(p, i, j, k)
                    out[k, j, i] = F(arr[k, j, i] - F(hi - lo))
    return ox, oy, oz


def test_second_transcription_of_force_and_projection_agrees_with_the_oracle(oracle):
    from conftest import random_fields
    nx, ny, nz = 7, 5, 6
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, jacobi_iters=5, temperature_env=0.25))
    st = random_fields(nx, ny, nz, seed=99, vel_scale=2.0)
    for fid, arr in st.items():
        o.set(fid, arr)
    o.stage("force")
    uy = force(st[1], st[4], F(o.params.dt), F(0.25))
    assert np.array_equal(o.get(1), uy)
    o.stage("project")                     # divergence + 5 sweeps (the #if 0 branch of Solver::Project)
    div = divergence(st[0], uy, st[2])
    assert np.array_equal(o.get(6), div)
    p = st[5]
    for _ in range(5):
        p = jacobi(p, div)
    assert np.array_equal(o.get(5), p)
    o.stage("gradient")
    gx, gy, gz = gradient(st[0], uy, st[2], p)
    assert np.array_equal(o.get(0), gx) and np.array_equal(o.get(1), gy) and np.array_equal(o.get(2), gz)
    o.close()


class FlatView:
    """A vorticity buffer as the reference indexes it: one flat array read with the strides AND
    clamps of whatever dims the call site passes (sample(), :30-34).  The buffer is
    (nx+1)(ny+1)(nz+1) zero-initialised floats, the defined-behaviour variant of this build."""

    def __init__(self, flat, dims):
        self.flat, self.dims = flat, dims

    def sample(self, x, y, z):
        dx, dy, dz = self.dims
        x = max(0, min(x, dx - 1)); y = max(0, min(y, dy - 1)); z = max(0, min(z, dz - 1))
        return self.flat[x + y * dx + z * dx * dy]


def trilerp_view(view, pos, off):   # :52-113 with b_minmax_sampe = true on a FlatView
    ix, fx = split(F(pos[0] - off[0]))
    iy, fy = split(F(pos[1] - off[1]))
    iz, fz = split(F(pos[2] - off[2]))
    tap = lambda a, b, c: view.sample(ix + a, iy + b, iz + c)
    l1 = lerp(lerp(tap(0, 0, 0), tap(1, 0, 0), fx), lerp(tap(0, 1, 0), tap(1, 1, 0), fx), fy)
    l2 = lerp(lerp(tap(0, 0, 1), tap(1, 0, 1), fx), lerp(tap(0, 1, 1), tap(1, 1, 1), fx), fy)
    return lerp(l1, l2, fz)


def find_vort(ux, uy, uz, nx, ny, nz):   # FindVortx/y/z :485-525, launched <<<(ny, nz), nx>>> (:1035-1037)
    size = (nx + 1) * (ny + 1) * (nz + 1)
    vx, vy, vz = np.zeros(size, F), np.zeros(size, F), np.zeros(size, F)
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                ind = i + j * nx + k * nx * ny
                vx[ind] = F(F(F(sample(uy, i, j, k) - sample(uy, i, j, k - 1)) - sample(uz, i, j, k)) + sample(uz, i, j - 1, k))
                vy[ind] = F(F(F(sample(uz, i, j, k) - sample(uz, i - 1, j, k)) - sample(ux, i, j, k)) + sample(ux, i, j, k - 1))
                vz[ind] = F(F(F(sample(ux, i, j, k) - sample(ux, i, j - 1, k)) - sample(uy, i, j, k)) + sample(uy, i - 1, j, k))
    return vx, vy, vz


def confine(u, direction, vx, vy, vz, nx, ny, nz, coeff):   # ApplyVortConfinement :540-605
    out = u.copy()
    dz, dy, dx = u.shape                         # = gridDim.y, gridDim.x, blockDim.x of :1078-1080
    views = (FlatView(vx, (nx, ny + 1, nz + 1)), FlatView(vy, (nx + 1, ny, nz + 1)), FlatView(vz, (nx + 1, ny + 1, nz)))
    offs = ((F(.5), F(0), F(0)), (F(0), F(.5), F(0)), (F(0), F(0), F(.5)))
    o = OFF[direction]

    def vort(pos):
        return [trilerp_view(views[q], pos, offs[q]) for q in range(3)]

    def norm(pos):
        a, b, c = vort(pos)
        return F(np.sqrt(F(F(F(a * a) + F(b * b)) + F(c * c))))

    h = 0.5   # the literals 0.5 of :556-588 are doubles: the sum is rounded when it becomes a float
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                if not (2 < i < dx - 2 and 2 < j < dy - 2 and 2 < k < dz - 2):
                    continue
                px, py, pz = F(F(i) + o[0]), F(F(j) + o[1]), F(F(k) + o[2])
                vt, vbo = norm([px, F(float(py) + h), pz]), norm([px, F(float(py) - h), pz])
                vr, vl = norm([F(float(px) + h), py, pz]), norm([F(float(px) - h), py, pz])
                vf, vbh = norm([px, py, F(float(pz) + h)]), norm([px, py, F(float(pz) - h)])
                vxc, vyc, vzc = vort([px, py, pz])
                gx, gy, gz = F(vr - vl), F(vt - vbo), F(vf - vbh)
                s = F(F(F(gx * gx) + F(gy * gy)) + F(gz * gz))
                n = F(math.sqrt(float(s) + 1e-5))                  # :593, double sqrt, narrowed
                gx, gy, gz = F(gx / n), F(gy / n), F(gz / n)
                if direction == 1:
                    d = F(coeff * F(F(gy * vzc) - F(gz * vyc)))
                elif direction == 2:
                    d = F(coeff * F(F(gz * vxc) - F(gx * vzc)))
                else:
                    d = F(coeff * F(F(gx * vyc) - F(gy * vxc)))
                out[k, j, i] = F(u[k, j, i] + d)
    return out


def test_second_transcription_of_the_vorticity_confinement_agrees_with_the_oracle(oracle):
    from conftest import random_fields
    nx, ny, nz = 9, 8, 10
    o = oracle.Oracle(oracle.default_params(nx, ny, nz, vort_confine=1))
    st = random_fields(nx, ny, nz, seed=123, vel_scale=1.0)
    for fid, arr in st.items():
        o.set(fid, arr)
    o.stage("vorticity")
    o.stage("confine")
    vx, vy, vz = find_vort(st[0], st[1], st[2], nx, ny, nz)
    c = F(o.params.curl_strength)
    for fid, d in ((0, 1), (1, 2), (2, 3)):
        want = confine(st[fid], d, vx, vy, vz, nx, ny, nz, c)
        got = o.get(fid)
        assert np.array_equal(got, want), "component %d: %d differ" % (fid, int((got != want).sum()))
        assert not np.array_equal(want, st[fid])
    o.close()


def cg_flat(f, off, i, j, k, d):   # cg_sample :37-42 on a flat level array
    dx, dy, dz = d
    if i < 0 or i >= dx or j < 0 or j >= dy or k < 0 or k >= dz:
        return F(0)
    return f[off + i + j * dx + k * dx * dy]


def neibor_sum(f, off, i, j, k, d):   # :44-50, summed left to right
    s = F(cg_flat(f, off, i - 1, j, k, d) + cg_flat(f, off, i + 1, j, k, d))
    s = F(s + cg_flat(f, off, i, j - 1, k, d))
    s = F(s + cg_flat(f, off, i, j + 1, k, d))
    s = F(s + cg_flat(f, off, i, j, k - 1, d))
    return F(s + cg_flat(f, off, i, j, k + 1, d))


def smooth(r, z, off, phase, d):   # Smooth :848-860: cells with (i+j+k) & 1 == phase, in place
    dx, dy, dz = d
    new = {}
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                if ((i + j + k) & 1) == phase:
                    ind = off + i + j * dx + k * dx * dy
                    new[ind] = F(F(r[ind] + neibor_sum(z, off, i, j, k, d)) / F(6))
    for ind, v in new.items():   # one colour only reads the other: order within a launch is immaterial
        z[ind] = v


def restrict(r, z, off, d):   # Restrict :862-877 = SubRestrict :822-834 for phases 0..7 in that order
    dx, dy, dz = d
    n = dx * dy * dz
    for phase in range(8):
        for ck in range(dz // 2):
            for cj in range(dy // 2):
                for ci in range(dx // 2):
                    i, j, k = ci * 2 + (phase & 1), cj * 2 + ((phase >> 1) & 1), ck * 2 + ((phase >> 2) & 1)
                    ind = off + i + j * dx + k * dx * dy
                    res = F(r[ind] - F(F(F(6) * z[ind]) - neibor_sum(z, off, i, j, k, d)))
                    new_ind = off + n + ci + cj * (dx // 2) + ck * (dx // 2) * (dy // 2)
                    r[new_ind] = F(float(r[new_ind]) + float(res) * 0.5)    # float += float * double


def prolongate(z, off, d):   # :836-846
    dx, dy, dz = d
    n = dx * dy * dz
    for k in range(dz):
        for j in range(dy):
            for i in range(dx):
                c = off + n + (i >> 1) + (j >> 1) * (dx >> 1) + (k >> 1) * (dx >> 1) * (dy >> 1)
                ind = off + i + j * dx + k * dx * dy
                z[ind] = F(z[ind] + z[c])


def v_cycle(r0, dims, levels, init_steps, bottom_steps):   # MG_Preconditioner :1312-1366
    nx, ny, nz = dims
    space, t = 0, nx * ny * nz
    for _ in range(levels):
        space += t
        t //= 8
    r, z = np.zeros(space, F), np.zeros(space, F)   # "initialize z[l] and r[l] with 0 except r[0]"
    r[:nx * ny * nz] = r0.reshape(-1)
    d, off = (nx, ny, nz), 0
    for l in range(levels - 1):
        for _ in range(init_steps << l):
            smooth(r, z, off, 0, d)
            smooth(r, z, off, 1, d)
        restrict(r, z, off, d)
        off += d[0] * d[1] * d[2]
        d = (d[0] >> 1, d[1] >> 1, d[2] >> 1)
    for _ in range(bottom_steps):
        smooth(r, z, off, 0, d)
        smooth(r, z, off, 1, d)
    for l in range(levels - 2, -1, -1):
        d = (d[0] << 1, d[1] << 1, d[2] << 1)
        off -= d[0] * d[1] * d[2]
        prolongate(z, off, d)
        for _ in range(init_steps << l):
            smooth(r, z, off, 0, d)
            smooth(r, z, off, 1, d)
    return r, z


def test_second_transcription_of_the_multigrid_v_cycle_agrees_with_the_oracle(oracle):
    rng = np.random.default_rng(8)
    dims = (8, 16, 8)
    r0 = rng.standard_normal((dims[2], dims[1], dims[0])).astype(F)
    _z0, r_o, z_o = oracle.mg_precondition(r0, dims, levels=3, init_smooth_steps=2, bottom_smooth_steps=5)
    r_t, z_t = v_cycle(r0, dims, 3, 2, 5)
    assert np.array_equal(r_o, r_t), "restricted residuals: %d differ" % int((r_o != r_t).sum())
    assert np.array_equal(z_o, z_t), "corrections: %d differ" % int((z_o != z_t).sum())
    # ComputeAp :789-799
    p = rng.standard_normal((4, 5, 6)).astype(F)
    want = np.empty_like(p)
    flat = p.reshape(-1)
    for k in range(4):
        for j in range(5):
            for i in range(6):
                want[k, j, i] = F(F(F(6) * cg_flat(flat, 0, i, j, k, (6, 5, 4))) - neibor_sum(flat, 0, i, j, k, (6, 5, 4)))
    assert np.array_equal(oracle.apply_a(p), want)
