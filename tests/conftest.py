import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def fs():
    import fluidsolver_amd as m
    m.load_library()
    return m


def random_fields(nx, ny, nz, seed, vel_scale=3.0):
    """Seeded random initial state, dense (z, y, x) float32 arrays keyed by field id 0..6."""
    rng = np.random.default_rng(seed)
    def r(shape, scale):
        return (rng.standard_normal(shape) * scale).astype(np.float32)
    return {
        0: r((nz, ny, nx + 1), vel_scale),
        1: r((nz, ny + 1, nx), vel_scale),
        2: r((nz + 1, ny, nx), vel_scale),
        3: rng.random((nz, ny, nx), dtype=np.float32),
        4: r((nz, ny, nx), 1.0),
        5: r((nz, ny, nx), 1.0),
        6: r((nz, ny, nx), 1.0),
    }


def smooth_velocity(nx, ny, nz, amp=2.5):
    """A smooth, divergent swirl: exercises in-range back-traces and wall crossings."""
    z, y, x = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    fx, fy, fz = 2 * np.pi / max(nx, 2), 2 * np.pi / max(ny, 2), 2 * np.pi / max(nz, 2)
    ux = amp * np.sin(fy * y) * np.cos(fz * z) + 0.3 * amp * np.cos(fx * x)
    uy = amp * np.sin(fz * z) * np.cos(fx * x) + 0.5
    uz = amp * np.sin(fx * x) * np.cos(fy * y) - 0.25
    return (ux[:nz, :ny, :].astype(np.float32), uy[:nz, :, :nx].astype(np.float32),
            uz[:, :ny, :nx].astype(np.float32))
