"""fluidsolver_amd -- host-side Python mirror of the reference's ``Solver`` interface
(CuSolver/Include/Solver.h:13-44) over the C ABI of ``include/fluidsolver.h``.

The GPU library ``fluidsolver_amd/lib/libfluidsolver.so`` (hand-written gfx950 HIP
kernels) is the only compute path: there is no CPU fallback, and importing
``Solver`` without the built library raises.
"""
from ._capi import (  # noqa: F401
    FIELD_NAMES,
    FS_DIV,
    FS_FLAG_FMAD,
    FS_FLAG_JACOBI_NAIVE,
    FS_FLAG_NO_GRAPH,
    FS_FLAG_TEST_SELF_PEER,
    FS_FLAG_VORT_CONFINE,
    FS_P,
    FS_PRESSURE_JACOBI,
    FS_PRESSURE_MGPCG,
    FS_RHO,
    FS_SCHEME_MACCORMACK,
    FS_SCHEME_RK4,
    FS_T,
    FS_UX,
    FS_UY,
    FS_UZ,
    FsError,
    Params,
    Solver,
    default_params,
    field_shape,
    lib_path,
    load_library,
)

__all__ = [
    "Solver", "Params", "default_params", "FsError", "field_shape", "load_library", "lib_path",
    "FS_UX", "FS_UY", "FS_UZ", "FS_RHO", "FS_T", "FS_P", "FS_DIV", "FIELD_NAMES",
    "FS_SCHEME_RK4", "FS_SCHEME_MACCORMACK", "FS_FLAG_NO_GRAPH", "FS_FLAG_JACOBI_NAIVE",
    "FS_FLAG_VORT_CONFINE", "FS_FLAG_TEST_SELF_PEER", "FS_FLAG_FMAD", "FS_PRESSURE_JACOBI", "FS_PRESSURE_MGPCG",
]
