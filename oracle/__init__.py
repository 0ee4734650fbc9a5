"""ctypes binding of the CPU oracle (oracle/fs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The shipped package ``fluidsolver_amd`` never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfs_oracle.so")

SCHEME_RK4 = 0
SCHEME_MACCORMACK = 1
UX, UY, UZ, RHO, T, P, DIV, R, VORTX, VORTY, VORTZ = range(11)
FIELD_NAMES = {UX: "ux", UY: "uy", UZ: "uz", RHO: "rho", T: "T", P: "p", DIV: "div", R: "r"}


class Params(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("dt", C.c_float),
        ("jacobi_iters", C.c_int),
        ("scheme", C.c_int),
        ("source_x", C.c_float), ("source_y", C.c_float), ("source_z", C.c_float),
        ("source_radius", C.c_float),
        ("rho0", C.c_float), ("temperature0", C.c_float), ("temperature_env", C.c_float),
        ("vort_confine", C.c_int), ("curl_strength", C.c_float),
        ("pressure_solver", C.c_int), ("mg_level", C.c_int), ("init_smooth_steps", C.c_int),
        ("bottom_smooth_steps", C.c_int), ("iter_precision", C.c_float),
    ]


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc, no FMA contraction)."""
    src = os.path.join(_HERE, "fs_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and (not os.path.exists(src) or os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.fso_set_fmad.argtypes = [C.c_int]
    L.fso_default_params.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int]
    L.fso_create.argtypes = [C.POINTER(Params)]
    L.fso_create.restype = C.c_void_p
    for name in ("fso_destroy", "fso_step", "fso_stage_source", "fso_stage_force",
                 "fso_stage_advect", "fso_stage_project", "fso_stage_gradient",
                 "fso_stage_vorticity", "fso_stage_confine"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = None
    L.fso_field.argtypes = [C.c_void_p, C.c_int]
    L.fso_field.restype = _fp
    L.fso_field_size.argtypes = [C.c_void_p, C.c_int]
    L.fso_field_size.restype = C.c_size_t
    L.fso_residual.argtypes = [C.c_void_p]
    L.fso_residual.restype = C.c_double
    L.fso_frame.argtypes = [C.c_void_p]
    L.fso_frame.restype = C.c_int
    L.fso_trilerp.argtypes = [_fp] + [C.c_float] * 6 + [C.c_int] * 4
    L.fso_trilerp.restype = C.c_float
    L.fso_backtrace.argtypes = [_fp, _fp, _fp] + [C.c_float] * 4 + [C.c_int] * 4 + [_fp]
    L.fso_backtrace.restype = None
    L.fso_source.argtypes = [C.c_float] * 4 + [_fp, _fp, C.c_float, C.c_float] + [C.c_int] * 3
    L.fso_source.restype = None
    L.fso_force_uy.argtypes = [_fp, _fp, C.c_float, C.c_float] + [C.c_int] * 3
    L.fso_force_uy.restype = None
    L.fso_semilag.argtypes = [_fp] * 5 + [C.c_float] + [C.c_int] * 5
    L.fso_semilag.restype = None
    L.fso_divergence.argtypes = [_fp] * 4 + [C.c_int] * 3
    L.fso_divergence.restype = None
    L.fso_jacobi.argtypes = [_fp] * 4 + [C.c_int] * 3
    L.fso_jacobi.restype = None
    L.fso_gradient.argtypes = [_fp] * 4 + [C.c_int] * 3
    L.fso_gradient.restype = None
    L.fso_sum_sq.argtypes = [_fp, C.c_size_t]
    L.fso_sum_sq.restype = C.c_double
    L.fso_last_cg_iters.argtypes = [C.c_void_p]
    L.fso_last_cg_iters.restype = C.c_int
    L.fso_last_rtr.argtypes = [C.c_void_p]
    L.fso_last_rtr.restype = C.c_float
    L.fso_mg_space.argtypes = [C.c_int] * 4
    L.fso_mg_space.restype = C.c_size_t
    L.fso_mg_smooth.argtypes = [_fp, _fp, C.c_size_t] + [C.c_int] * 4
    L.fso_mg_smooth.restype = None
    L.fso_mg_restrict.argtypes = [_fp, _fp, C.c_size_t] + [C.c_int] * 3
    L.fso_mg_restrict.restype = None
    L.fso_mg_prolongate.argtypes = [_fp, C.c_size_t] + [C.c_int] * 3
    L.fso_mg_prolongate.restype = None
    L.fso_mg_precondition.argtypes = [_fp, _fp] + [C.c_int] * 6
    L.fso_mg_precondition.restype = None
    L.fso_dot.argtypes = [_fp, _fp, C.c_size_t]
    L.fso_dot.restype = C.c_float
    L.fso_apply_a.argtypes = [_fp, _fp] + [C.c_int] * 3
    L.fso_apply_a.restype = None
    L.fso_vort_size.argtypes = [C.c_int] * 3
    L.fso_vort_size.restype = C.c_size_t
    L.fso_find_vort.argtypes = [_fp] * 6 + [C.c_int] * 3
    L.fso_find_vort.restype = None
    L.fso_vort_confine.argtypes = [_fp] * 4 + [C.c_int] * 4 + [C.c_float]
    L.fso_vort_confine.restype = None
    _lib = L
    return L


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def field_shape(which, nx, ny, nz):
    """numpy shape (z, y, x) of a field; x is the fastest axis as in the reference."""
    if which == UX:
        return (nz, ny, nx + 1)
    if which == UY:
        return (nz, ny + 1, nx)
    if which == UZ:
        return (nz + 1, ny, nx)
    if which in (VORTX, VORTY, VORTZ):
        return ((nz + 1) * (ny + 1) * (nx + 1),)      # flat: written and read with different strides
    return (nz, ny, nx)


def set_fmad(on):
    """Process-wide contraction switch of the oracle (see fso_set_fmad); always reset it to False."""
    lib().fso_set_fmad(1 if on else 0)


def default_params(nx, ny, nz, **kw):
    p = Params()
    lib().fso_default_params(C.byref(p), nx, ny, nz)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Oracle:
    """Stateful whole-step oracle; fields are numpy views into the C state."""

    def __init__(self, params):
        self.params = params
        self._L = lib()
        self._s = self._L.fso_create(C.byref(params))
        if not self._s:
            raise MemoryError("fso_create failed")

    def close(self):
        if self._s:
            self._L.fso_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, n=1):
        for _ in range(n):
            self._L.fso_step(self._s)

    def stage(self, name):
        getattr(self._L, "fso_stage_" + name)(self._s)

    def field(self, which):
        """Live view (valid until the next step swaps buffers)."""
        p = self.params
        ptr = self._L.fso_field(self._s, which)
        shape = field_shape(which, p.nx, p.ny, p.nz)
        return np.ctypeslib.as_array(ptr, shape=shape)

    def get(self, which):
        return self.field(which).copy()

    def set(self, which, arr):
        self.field(which)[...] = np.asarray(arr, dtype=np.float32).reshape(self.field(which).shape)

    def residual(self):
        return self._L.fso_residual(self._s)

    def cg_info(self):
        """(iterations, rTr) of the last MGPCG solve."""
        return self._L.fso_last_cg_iters(self._s), self._L.fso_last_rtr(self._s)


# ---- stateless kernel-level wrappers (arrays are (z, y, x) float32, C-contiguous) ----

def trilerp(field, pos, offs, clamp):
    dz, dy, dx = field.shape
    return lib().fso_trilerp(_p(field), pos[0], pos[1], pos[2], offs[0], offs[1], offs[2],
                             dx, dy, dz, int(clamp))


def backtrace(ux, uy, uz, pos, dt, scheme=SCHEME_MACCORMACK):
    nz, ny, nx = ux.shape[0], ux.shape[1], ux.shape[2] - 1
    out = np.zeros(3, np.float32)
    lib().fso_backtrace(_p(ux), _p(uy), _p(uz), pos[0], pos[1], pos[2], dt, nx, ny, nz, scheme, _p(out))
    return out


def source(rho, temp, sx, sy, sz, radius, rho0, t0):
    nz, ny, nx = rho.shape
    lib().fso_source(sx, sy, sz, radius, _p(rho), _p(temp), rho0, t0, nx, ny, nz)


def force_uy(uy, temp, dt, t_env):
    nz, ny, nx = temp.shape
    lib().fso_force_uy(_p(uy), _p(temp), dt, t_env, nx, ny, nz)


def semilag(field, ux, uy, uz, dt, dir_, scheme=SCHEME_MACCORMACK):
    nz, ny, nx = ux.shape[0], ux.shape[1], ux.shape[2] - 1
    out = np.empty_like(field)
    lib().fso_semilag(_p(field), _p(out), _p(ux), _p(uy), _p(uz), dt, nx, ny, nz, dir_, scheme)
    return out


def divergence(ux, uy, uz):
    nz, ny, nx = ux.shape[0], ux.shape[1], ux.shape[2] - 1
    out = np.empty((nz, ny, nx), np.float32)
    lib().fso_divergence(_p(out), _p(ux), _p(uy), _p(uz), nx, ny, nz)
    return out


def jacobi(p, div, want_r=True):
    nz, ny, nx = p.shape
    newp = np.empty_like(p)
    r = np.empty_like(p) if want_r else None
    lib().fso_jacobi(_p(p), _p(newp), _p(div), _p(r) if want_r else None, nx, ny, nz)
    return newp, r


def gradient(ux, uy, uz, p):
    nz, ny, nx = p.shape
    lib().fso_gradient(_p(ux), _p(uy), _p(uz), _p(p), nx, ny, nz)


def sum_sq(r):
    return lib().fso_sum_sq(_p(r), r.size)


def vort_size(nx, ny, nz):
    return lib().fso_vort_size(nx, ny, nz)


def find_vort(ux, uy, uz):
    """The three vorticity arrays (flat, (nx+1)(ny+1)(nz+1) floats each, zero beyond nx*ny*nz)."""
    nz, ny, nx = ux.shape[0], ux.shape[1], ux.shape[2] - 1
    n = vort_size(nx, ny, nz)
    vx, vy, vz = (np.zeros(n, np.float32) for _ in range(3))
    lib().fso_find_vort(_p(vx), _p(vy), _p(vz), _p(ux), _p(uy), _p(uz), nx, ny, nz)
    return vx, vy, vz


def vort_confine(u, vx, vy, vz, dims, dir_, coeff):
    """In-place confinement force on one velocity component (dir_ 1, 2, 3)."""
    nx, ny, nz = dims
    lib().fso_vort_confine(_p(u), _p(vx), _p(vy), _p(vz), nx, ny, nz, dir_, coeff)


PRESSURE_JACOBI, PRESSURE_MGPCG = 0, 1


def mg_space(nx, ny, nz, levels=4):
    return lib().fso_mg_space(nx, ny, nz, levels)


def mg_precondition(r0, dims, levels=4, init_smooth_steps=2, bottom_smooth_steps=50):
    """One V-cycle on the level-0 right-hand side r0 (shape (nz, ny, nx)); returns z of level 0."""
    nx, ny, nz = dims
    space = mg_space(nx, ny, nz, levels)
    r = np.zeros(space, np.float32)
    z = np.zeros(space, np.float32)
    r[:nx * ny * nz] = np.asarray(r0, np.float32).reshape(-1)
    lib().fso_mg_precondition(_p(r), _p(z), nx, ny, nz, levels, init_smooth_steps, bottom_smooth_steps)
    return z[:nx * ny * nz].reshape(nz, ny, nx).copy(), r, z


def apply_a(p):
    nz, ny, nx = p.shape
    out = np.empty_like(p)
    lib().fso_apply_a(_p(out), _p(p), nx, ny, nz)
    return out


def dot(a, b):
    return lib().fso_dot(_p(a.reshape(-1)), _p(b.reshape(-1)), a.size)
