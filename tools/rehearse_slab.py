#!/usr/bin/env python3
"""Rehearsal of one interior rank of the weak-scaling column on a single GPU.

The middle slab (n x n x n) of a n x n x 3n column runs alone and trades its boundary planes with
itself over RCCL (FS_FLAG_TEST_SELF_PEER), once with the transfers ordered through the communication
stream (the default of the RCCL transport) and once with everything on the compute stream.  The
transfers stay inside one GPU's memory, so this prices the slab overheads (boundary launches,
wider halos, eager launches) and the stream ordering, not xGMI.

    python tools/rehearse_slab.py [n] [sweeps] [steps]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    import fluidsolver_amd as fs
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)

    ref = fs.Solver(fs.default_params(n, n, n, jacobi_iters=sweeps)).Initialize()
    ref.step(3); ref.sync()
    t0 = time.perf_counter(); ref.step(steps); ref.sync()
    single = (time.perf_counter() - t0) / steps * 1e3
    ref.close()
    print("single domain %d^3, %d sweeps: %.2f ms/step" % (n, sweeps, single))
    for streams, name in (("1", "two streams (RCCL default)"), ("0", "one stream")):
        os.environ["FS_XCHG_STREAMS"] = streams
        p = fs.default_params(n, n, 3 * n, jacobi_iters=sweeps, source_z=1.5 * n, flags=fs.FS_FLAG_TEST_SELF_PEER)
        s = slabmod.SlabSolver(p, 1, 3, transport="rccl")
        uid = (C.c_char * slabmod.FS_UNIQUE_ID_BYTES)()
        s._check(L.fs_comm_unique_id(uid))
        h = C.c_void_p()
        s._check(L.fs_create_slab(C.byref(p), 1, 3, uid, slabmod._EXCHANGE_FN(0), None, C.byref(h)))
        s._h = h
        s.step(3); s.sync()
        t0 = time.perf_counter(); s.step(steps); s.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print("interior slab, %-28s %.2f ms/step  (%.0f %% of the single-domain rate)" % (name + ":", ms, 100 * single / ms))
        s.close()


if __name__ == "__main__":
    main()
