"""ctypes binding of include/fluidsolver.h.

Every call goes through the C ABI exactly as a C/C++ host would; arrays cross the
boundary as plain float pointers in the reference's dense x-fastest layout.
"""
import ctypes as C
import os

import numpy as np

FS_UX, FS_UY, FS_UZ, FS_RHO, FS_T, FS_P, FS_DIV = range(7)
FIELD_NAMES = {FS_UX: "ux", FS_UY: "uy", FS_UZ: "uz", FS_RHO: "rho", FS_T: "T", FS_P: "p", FS_DIV: "div"}
FS_SCHEME_RK4, FS_SCHEME_MACCORMACK = 0, 1
FS_PRESSURE_JACOBI, FS_PRESSURE_MGPCG = 0, 1
FS_FLAG_NO_GRAPH = 1 << 0
FS_FLAG_JACOBI_NAIVE = 1 << 1
FS_FLAG_VORT_CONFINE = 1 << 2
FS_FLAG_TEST_SELF_PEER = 1 << 3
FS_FLAG_FMAD = 1 << 4

_HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/fluidsolver.h declares
EXPORTED_SYMBOLS = (
    "fs_default_params", "fs_create", "fs_destroy", "fs_step", "fs_steps", "fs_sync", "fs_frame", "fs_step_graphs",
    "fs_stage_source", "fs_stage_force", "fs_stage_vorticity", "fs_stage_advect", "fs_stage_confine",
    "fs_stage_divergence",
    "fs_stage_jacobi", "fs_stage_project", "fs_stage_pressure_solve", "fs_mg_vcycle", "fs_stage_gradient", "fs_density_host",
    "fs_velocity_host", "fs_field_host", "fs_field_host_begin", "fs_field_host_end", "fs_field_size", "fs_field_upload", "fs_field_fill", "fs_field_device",
    "fs_residual", "fs_residual_noalloc", "fs_last_sweeps", "fs_cg_residual", "fs_last_error", "fs_version",
    "fs_profile_enable", "fs_profile_read", "fs_profile_reset", "fs_halo_wait",
    "fs_comm_unique_id", "fs_create_slab", "fs_slab_info", "fs_comm_ranks", "fs_jacobi_config", "fs_jacobi_deep_config",
)

STAGES = ("source", "force", "advect_velocity", "advect_scalars", "divergence", "jacobi", "gradient",
          "vorticity", "confine", "mgpcg")


class FsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fluidsolver error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """struct fs_params"""
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("dt", C.c_float),
        ("jacobi_iters", C.c_int),
        ("scheme", C.c_int),
        ("source_x", C.c_float), ("source_y", C.c_float), ("source_z", C.c_float),
        ("source_radius", C.c_float),
        ("source_rho", C.c_float), ("source_temperature", C.c_float),
        ("temperature_env", C.c_float),
        ("device", C.c_int),
        ("flags", C.c_uint),
        ("tolerance", C.c_float),
        ("check_every", C.c_int),
        ("max_sweeps", C.c_int),
        ("curl_strength", C.c_float),
        ("pressure_solver", C.c_int), ("mg_level", C.c_int), ("init_smooth_steps", C.c_int),
        ("bottom_smooth_steps", C.c_int), ("iter_precision", C.c_float),
    ]


class Layout(C.Structure):
    """struct fs_layout"""
    _fields_ = [
        ("dx", C.c_int), ("dy", C.c_int), ("dz", C.c_int),
        ("z0", C.c_int), ("nz_local", C.c_int), ("ghost", C.c_int),
        ("pitch", C.c_size_t), ("plane", C.c_size_t),
    ]


def lib_path():
    # FS_LIB_PATH: an alternative build of the same library (A/B timing of kernel variants)
    return os.environ.get("FS_LIB_PATH") or os.path.join(_HERE, "lib", "libfluidsolver.so")


_lib = None
_fp = C.POINTER(C.c_float)


def load_library():
    """Load the HIP library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "fluidsolver_amd: %s is missing -- build it with `make` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.fs_default_params.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int]
    L.fs_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    for name in ("fs_destroy", "fs_step", "fs_sync", "fs_frame", "fs_stage_source", "fs_stage_force",
                 "fs_stage_advect", "fs_stage_divergence", "fs_stage_project", "fs_stage_gradient",
                 "fs_stage_vorticity", "fs_stage_confine", "fs_last_sweeps", "fs_step_graphs",
                 "fs_stage_pressure_solve", "fs_mg_vcycle"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_int
    L.fs_steps.argtypes = [vp, C.c_int]
    L.fs_stage_jacobi.argtypes = [vp, C.c_int]
    L.fs_density_host.argtypes = [vp]
    L.fs_density_host.restype = _fp
    L.fs_velocity_host.argtypes = [vp, C.c_int]
    L.fs_velocity_host.restype = _fp
    L.fs_field_host.argtypes = [vp, C.c_int]
    L.fs_field_host.restype = _fp
    L.fs_field_size.argtypes = [vp, C.c_int]
    L.fs_field_host_begin.argtypes = [vp, C.c_int]
    L.fs_field_host_end.argtypes = [vp, C.c_int]
    L.fs_field_host_end.restype = _fp
    L.fs_field_size.restype = C.c_size_t
    L.fs_field_upload.argtypes = [vp, C.c_int, _fp]
    L.fs_field_fill.argtypes = [vp, C.c_int, C.c_float]
    L.fs_field_device.argtypes = [vp, C.c_int, C.POINTER(_fp), C.POINTER(Layout)]
    L.fs_residual.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.fs_residual_noalloc.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.fs_comm_ranks.argtypes = [vp]
    L.fs_jacobi_config.argtypes = [vp] + [C.POINTER(C.c_int)] * 5
    L.fs_jacobi_deep_config.argtypes = [vp] + [C.POINTER(C.c_int)] * 3
    L.fs_cg_residual.argtypes = [vp, C.POINTER(C.c_float)]
    L.fs_profile_enable.argtypes = [vp, C.c_int]
    L.fs_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.fs_profile_reset.argtypes = [vp]
    L.fs_halo_wait.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.fs_last_error.restype = C.c_char_p
    L.fs_version.restype = C.c_char_p
    _lib = L
    return L


def field_shape(field, nx, ny, nz):
    """numpy shape (z, y, x) of a dense field."""
    if field == FS_UX:
        return (nz, ny, nx + 1)
    if field == FS_UY:
        return (nz, ny + 1, nx)
    if field == FS_UZ:
        return (nz + 1, ny, nx)
    return (nz, ny, nx)


def default_params(nx, ny, nz, **kw):
    """Reference defaults (Solver.h:58-81) for an nx x ny x nz grid, overridden by kw."""
    L = load_library()
    p = Params()
    rc = L.fs_default_params(C.byref(p), nx, ny, nz)
    if rc:
        raise FsError(rc, L.fs_last_error().decode())
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError("fs_params has no member %r" % k)
        setattr(p, k, v)
    return p


class Solver:
    """Mirror of the reference's ``class Solver`` (Solver.h:13-44).

    ``Initialize`` / ``Update`` / ``GetDensityField`` keep the reference names and
    meaning; the snake_case methods expose the rest of the C ABI.
    """

    def __init__(self, params=None, nx=64, ny=128, nz=64, **kw):
        self._L = load_library()
        self.params = params if params is not None else default_params(nx, ny, nz, **kw)
        self._h = None

    # -- reference interface ---------------------------------------------------
    def Initialize(self):
        if self._h is not None:
            raise FsError(-6, "Initialize called twice")
        h = C.c_void_p()
        self._check(self._L.fs_create(C.byref(self.params), C.byref(h)))
        self._h = h
        return self

    def Update(self):
        self._check(self._L.fs_step(self._handle()))

    def GetDensityField(self):
        return self.field(FS_RHO)

    def Advect(self):
        self._check(self._L.fs_stage_advect(self._handle()))

    def Project(self):
        self._check(self._L.fs_stage_project(self._handle()))

    # -- the rest of the C ABI -------------------------------------------------
    def _handle(self):
        if self._h is None:
            raise FsError(-6, "solver not initialised")
        return self._h

    def _check(self, rc):
        if rc:
            raise FsError(rc, self._L.fs_last_error().decode())

    def close(self):
        if self._h is not None:
            self._L.fs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        if self._h is None:
            self.Initialize()
        return self

    def __exit__(self, *a):
        self.close()

    def step(self, n=1):
        self._check(self._L.fs_steps(self._handle(), n))

    def sync(self):
        self._check(self._L.fs_sync(self._handle()))

    @property
    def frame(self):
        return self._L.fs_frame(self._handle())

    def stage(self, name, *args):
        self._check(getattr(self._L, "fs_stage_" + name)(self._handle(), *args))

    def field_begin(self, field):
        """Start an asynchronous read-back of a field (snapshot now, transfer overlaps later steps)."""
        self._check(self._L.fs_field_host_begin(self._handle(), field))

    def local_shape(self, field):
        """numpy shape (z, y, x) of what the host getters return for a field (a z-slab overrides
        this: the getters move its own planes only)."""
        p = self.params
        return field_shape(field, p.nx, p.ny, p.nz)

    def field_end(self, field):
        ptr = self._L.fs_field_host_end(self._handle(), field)
        if not ptr:
            raise FsError(-6, self._L.fs_last_error().decode())
        return np.ctypeslib.as_array(ptr, shape=self.local_shape(field)).copy()

    def field(self, field):
        """Dense host copy (z, y, x) of a field; synchronises."""
        h = self._handle()
        ptr = self._L.fs_field_host(h, field)
        if not ptr:
            raise FsError(-2, self._L.fs_last_error().decode())
        return np.ctypeslib.as_array(ptr, shape=self.local_shape(field)).copy()

    def velocity(self, component):
        h = self._handle()
        ptr = self._L.fs_velocity_host(h, component)
        if not ptr:
            raise FsError(-2, self._L.fs_last_error().decode())
        return np.ctypeslib.as_array(ptr, shape=self.local_shape(FS_UX + component)).copy()

    def upload(self, field, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        if a.shape != self.local_shape(field):
            raise ValueError("field %d wants shape %s, got %s" % (field, self.local_shape(field), a.shape))
        self._check(self._L.fs_field_upload(self._handle(), field, a.ctypes.data_as(_fp)))

    def fill(self, field, value):
        self._check(self._L.fs_field_fill(self._handle(), field, value))

    def device_field(self, field):
        ptr = _fp()
        lay = Layout()
        self._check(self._L.fs_field_device(self._handle(), field, C.byref(ptr), C.byref(lay)))
        return C.cast(ptr, C.c_void_p).value, lay

    def residual(self, which=0):
        out = C.c_double()
        self._check(self._L.fs_residual(self._handle(), which, C.byref(out)))
        return out.value

    def residual_noalloc(self, which=0):
        """fs_residual through the library's static scratch array (no device allocation)."""
        out = C.c_double()
        self._check(self._L.fs_residual_noalloc(self._handle(), which, C.byref(out)))
        return out.value

    def jacobi_config(self):
        """{variant, tile_rows, aligned_rows, planes_per_chunk, nontemporal_stores} of the blocked
        Jacobi passes."""
        v = [C.c_int() for _ in range(5)]
        self._check(self._L.fs_jacobi_config(self._handle(), *[C.byref(x) for x in v]))
        d = dict(zip(("variant", "tile_rows", "aligned_rows", "planes_per_chunk", "nontemporal_stores"),
                     (x.value for x in v)))
        w = [C.c_int() for _ in range(3)]
        self._check(self._L.fs_jacobi_deep_config(self._handle(), *[C.byref(x) for x in w]))
        d.update(zip(("sweeps_per_pass", "deep_planes_per_chunk", "deep_nontemporal_stores"), (x.value for x in w)))
        return d

    @property
    def step_graphs(self):
        """captured step graphs held"""
        return self._L.fs_step_graphs(self._handle())

    @property
    def comm_ranks(self):
        return self._L.fs_comm_ranks(self._handle())

    def profile(self, on=True):
        self._check(self._L.fs_profile_enable(self._handle(), int(on)))

    def profile_reset(self):
        self._check(self._L.fs_profile_reset(self._handle()))

    def profile_read(self):
        """{stage: (device ms total, kernel launches)} since the last reset; synchronises."""
        out = {}
        for q, name in enumerate(STAGES):
            ms, n = C.c_double(), C.c_longlong()
            self._check(self._L.fs_profile_read(self._handle(), q, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def halo_wait(self):
        """{host_ms, device_ms, exchanges} a z-slab spent waiting for halos since the last
        profile_reset (zeros on one GPU); synchronises."""
        h, d, n = C.c_double(), C.c_double(), C.c_longlong()
        self._check(self._L.fs_halo_wait(self._handle(), C.byref(h), C.byref(d), C.byref(n)))
        return {"host_ms": h.value, "device_ms": d.value, "exchanges": n.value}

    @property
    def last_sweeps(self):
        return self._L.fs_last_sweeps(self._handle())

    def cg_info(self):
        """(iterations, r.r) of the last MGPCG solve."""
        rtr = C.c_float()
        self._check(self._L.fs_cg_residual(self._handle(), C.byref(rtr)))
        return self._L.fs_last_sweeps(self._handle()), rtr.value
