"""Raw field dumps: one little-endian float32 file per field, x fastest (the reference's dense
layout, flat index x + y*dx + z*dx*dy, CuSolver/Source/Solver.cu:33), plus a JSON sidecar with
the dims and the parameters of the run.  The reference has no file format; this one is shared by
the C++ bench (``fs_bench ... dump_prefix``), the Python binding and the parity tooling."""
import json
import os

import numpy as np

FORMAT = "fluidsolver-f32-v1"
_NAMES = {0: "ux", 1: "uy", 2: "uz", 3: "rho", 4: "T", 5: "p", 6: "div"}


def write_dump(solver, prefix, fields=(0, 1, 2, 3)):
    """Write the given fields of a ``fluidsolver_amd.Solver`` as <prefix>_<name>.f32 + <prefix>.json."""
    p = solver.params
    meta = {"format": FORMAT, "endian": "little", "order": "x-fastest", "nx": p.nx, "ny": p.ny, "nz": p.nz,
            "frame": solver.frame, "dt": float(p.dt), "jacobi_iters": p.jacobi_iters, "scheme": p.scheme,
            "flags": int(p.flags), "fields": {}}
    for f in fields:
        a = np.ascontiguousarray(solver.field(f), dtype="<f4")
        path = "%s_%s.f32" % (prefix, _NAMES[f])
        a.tofile(path)
        meta["fields"][_NAMES[f]] = {"file": os.path.basename(path), "dims": [a.shape[2], a.shape[1], a.shape[0]]}
    with open(prefix + ".json", "w") as fh:
        json.dump(meta, fh)
        fh.write("\n")
    return meta


def read_dump(prefix):
    """Returns (meta, {name: array shaped (dz, dy, dx)})."""
    with open(prefix + ".json") as fh:
        meta = json.load(fh)
    if meta.get("format") != FORMAT:
        raise ValueError("%s.json: unknown dump format %r" % (prefix, meta.get("format")))
    out = {}
    here = os.path.dirname(os.path.abspath(prefix + ".json"))
    for name, d in meta["fields"].items():
        dx, dy, dz = d["dims"]
        a = np.fromfile(os.path.join(here, d["file"]), dtype="<f4")
        if a.size != dx * dy * dz:
            raise ValueError("%s: %d floats, expected %d" % (d["file"], a.size, dx * dy * dz))
        out[name] = a.reshape(dz, dy, dx)
    return meta, out
