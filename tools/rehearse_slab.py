#!/usr/bin/env python3
"""Rehearsal of one interior rank of a z-slab run on a single GPU.

The middle slab (n x n x nz_local) of a n x n x (3 nz_local) column runs alone and trades its
boundary planes with itself over RCCL (FS_FLAG_TEST_SELF_PEER), once with the transfers ordered
through the communication stream (the default) and once with everything on the compute stream.
The transfers stay inside one GPU's memory, so this prices the slab overheads (boundary launches,
wider halos, eager launches, host enqueue time) and the stream ordering, not xGMI.

    python tools/rehearse_slab.py [n] [sweeps] [steps] [nz_local] [mgpcg]

nz_local = n rehearses a weak-scaling rank; nz_local = n / 8 a strong-scaling rank of an 8-GPU run.
Per configuration it prints the time per step, the host time spent enqueueing a step (the call
returns before the GPU has finished), and the device time of the Jacobi stage per fused pass, i.e.
the budget an exchange has to hide in.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(solver, steps):
    """(ms per step with the GPU drained, ms per step the host spent enqueueing)"""
    solver.step(3)
    solver.sync()
    t0 = time.perf_counter()
    solver.step(steps)
    t1 = time.perf_counter()
    solver.sync()
    t2 = time.perf_counter()
    return (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    nzl = int(sys.argv[4]) if len(sys.argv) > 4 else n
    mgpcg = len(sys.argv) > 5 and sys.argv[5] == "mgpcg"    # the MGPCG pressure solver, `sweeps` = its iteration limit
    import fluidsolver_amd as fs
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)
    out = {"n": n, "nz_local": nzl, "sweeps": sweeps, "steps": steps}

    extra = dict(pressure_solver=fs.FS_PRESSURE_MGPCG) if mgpcg else {}
    stage = "mgpcg" if mgpcg else "jacobi"
    ref = fs.Solver(fs.default_params(n, n, nzl, jacobi_iters=sweeps, **extra)).Initialize()
    single, single_host = timed(ref, steps)
    ref.close()
    out["single_domain_ms"] = single
    out["single_domain_host_enqueue_ms"] = single_host
    print("single domain %dx%dx%d, %d sweeps: %.2f ms/step (host enqueue %.3f ms, graph replay)"
          % (n, n, nzl, sweeps, single, single_host))
    for streams, name in (("1", "two_streams"), ("0", "one_stream")):
        os.environ["FS_XCHG_STREAMS"] = streams
        p = fs.default_params(n, n, 3 * nzl, jacobi_iters=sweeps, source_z=1.5 * nzl,
                              flags=fs.FS_FLAG_TEST_SELF_PEER, **extra)
        s = slabmod.SlabSolver(p, 1, 3, transport="rccl")
        uid = (C.c_char * slabmod.FS_UNIQUE_ID_BYTES)()
        s._check(L.fs_comm_unique_id(uid))
        h = C.c_void_p()
        s._check(L.fs_create_slab(C.byref(p), 1, 3, uid, slabmod._EXCHANGE_FN(0), None, C.byref(h)))
        s._h = h
        ms, host = timed(s, steps)
        # device time of the Jacobi stage (events around the stage: boundary launches, interior
        # launches and the waits for the exchanges)
        s.profile(True)
        s.profile_reset()
        s.step(steps)
        s.sync()
        prof = s.profile_read()
        s.profile(False)
        jac_ms, jac_launches = prof[stage]
        passes = steps * ((sweeps + 3) // 4)
        out[name] = {"ms_per_step": ms, "host_enqueue_ms_per_step": host,
                     "rate_vs_single_domain": single / ms,
                     "jacobi_ms_per_step": jac_ms / steps, "jacobi_us_per_pass": 1e3 * jac_ms / passes,
                     "jacobi_launches_per_step": jac_launches / steps}
        if mgpcg:
            out[name]["cg_iterations_last_step"] = s.cg_info()[0]
        print("interior slab, %-12s %.2f ms/step (%.0f %% of the single-domain rate), host enqueue %.3f ms/step, "
              "Jacobi %.2f ms/step = %.1f us per fused pass incl. its exchange"
              % (name + ":", ms, 100 * single / ms, host, jac_ms / steps, 1e3 * jac_ms / passes))
        s.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
