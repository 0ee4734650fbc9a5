"""fs_residual_noalloc: the residual norm through the library's file-scope device scratch array
(no device allocation inside the call) against fs_residual, which uses the solver's own scratch
buffer.  Both run the same kernels on the same operands, so the two doubles must be identical;
the oracle's sum is the second check (1e-5 relative, the tolerance of the residual norm)."""
import ctypes as C

import numpy as np
import pytest

from conftest import random_fields

pytestmark = pytest.mark.gpu


def _loaded(fs, dims, seed, sweeps):
    s = fs.Solver(fs.default_params(*dims, jacobi_iters=sweeps)).Initialize()
    st = random_fields(*dims, seed=seed)
    s.upload(fs.FS_P, st[5])
    s.upload(fs.FS_DIV, st[6])
    return s, st


def test_residual_noalloc_two_calls_second_smaller(fs, oracle):
    """Two calls in a row on one GPU share the static array; the second grid is smaller, so the
    array still holds partial sums of the first call beyond what the second one writes."""
    big, st_big = _loaded(fs, (48, 40, 24), 3, 7)
    small, st_small = _loaded(fs, (20, 12, 8), 4, 7)
    got_big = big.residual_noalloc(0)
    got_small = small.residual_noalloc(0)
    assert got_big == big.residual(0)
    assert got_small == small.residual(0)
    for got, st in ((got_big, st_big), (got_small, st_small)):
        _, r = oracle.jacobi(st[5], st[6])
        want = oracle.sum_sq(r)
        assert abs(got - want) <= 1e-5 * want
    # after sweeps: residual of the current iterate and of the input of the last single sweep
    for s in (big, small):
        s.stage("jacobi", 7)        # 4 fused + 3 single sweeps: the previous iterate is held
        assert s.residual_noalloc(0) == s.residual(0)
        assert s.residual_noalloc(1) == s.residual(1)
    # the big one again after the small one
    assert big.residual_noalloc(0) == big.residual(0)
    big.close()
    small.close()


def test_residual_noalloc_limits(fs):
    """Grids with more (y, z) rows than the static array holds are rejected with a clear error, and
    so are a missing previous iterate and bad arguments; fs_residual still serves such a grid."""
    L = fs.load_library()
    s = fs.Solver(fs.default_params(8, 1032, 1024, jacobi_iters=4)).Initialize()   # 1 056 768 rows > 2^20
    out = C.c_double()
    rc = L.fs_residual_noalloc(s._handle(), 0, C.byref(out))
    assert rc == -1 and b"exceed" in L.fs_last_error()
    assert s.residual(0) == 0.0
    s.close()
    s = fs.Solver(fs.default_params(16, 16, 16, jacobi_iters=4)).Initialize()
    s.stage("jacobi", 4)            # one fused pass: the iterate before the last sweep is gone
    with pytest.raises(fs.FsError) as e:
        s.residual_noalloc(1)
    assert e.value.code == -6
    assert L.fs_residual_noalloc(s._handle(), 2, C.byref(out)) == -1
    assert L.fs_residual_noalloc(None, 0, C.byref(out)) == -1
    s.close()
