#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the full simulation step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--iters 80] [--flow plume|swirl]

One step = source -> buoyancy -> advect (ux, uy, uz, T, rho) -> divergence ->
`iters` Jacobi sweeps -> pressure-gradient subtraction, on synthetic input (all-zero
initial state, sphere source stamped every step; no RNG).  At N = 1 the workload is the
512^3 / 80-sweep configuration (the size whose Jacobi working set does not fit the
256 MiB Infinity Cache).  For N > 1 (launched by torch.distributed.run, one rank per
GPU) the grid is the weak-scaling column 512 x 512 x (512*N), z-slab decomposed.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel class of the
step, timed with HIP events on the solver's own stream inside the timed region;
`cpu_baseline` is the CPU oracle (a restatement of the reference's loops, there is no
CPU path upstream) timed on this host on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
# algorithmic (compulsory) bytes per cell, fp32: each kernel reads every input once and
# writes every output once
BYTES_PER_CELL = {
    "force": 12.0,             # read T, read+write uy
    "advect_velocity": 48.0,   # 3 x (3 reads + 1 write)
    "advect_scalars": 40.0,    # 2 x (4 reads + 1 write)
    "divergence": 16.0,        # 3 reads + 1 write
    "jacobi": 12.0,            # per sweep: read p, read div, write p'
    "gradient": 28.0,          # read p, read+write 3 velocity components
}


def step_bytes_per_cell(iters):
    return 144.0 + 12.0 * iters


def cpu_baseline(iters, sample_n):
    """Oracle timed on the host cores: one step of sample_n^3 with the same sweep count."""
    import oracle
    oracle.build()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    warm = oracle.Oracle(oracle.default_params(32, 32, 32, jacobi_iters=4))
    warm.step()
    warm.close()
    o = oracle.Oracle(oracle.default_params(sample_n, sample_n, sample_n, jacobi_iters=iters))
    o.step()   # first step also pays first-touch page faults: leave it untimed
    t0 = time.perf_counter()
    o.step()
    dt = time.perf_counter() - t0
    o.close()
    return {
        "value": sample_n ** 3 / dt,
        "unit": "cell-updates/s",
        "cores": threads,
        "kind": "port",
        "sample": "1 step of %d^3, %d Jacobi sweeps (the 2nd step from the zero state; the 1st, untimed, pays the page faults), gcc -O2 "
                  "-ffp-contract=off + OpenMP, %.2f s" % (sample_n, iters, dt),
    }


def measured_copy_bandwidth(torch, device):
    """Device-to-device copy of 1 GiB (1 read + 1 write per byte) on this GPU, GB/s: what a plain
    streaming kernel reaches here, to read the spec-peak fractions against."""
    try:
        n = 1 << 28
        x = torch.empty(n, dtype=torch.float32, device="cuda:%d" % device)
        y = torch.empty_like(x)
        x.zero_()
        for _ in range(2):
            y.copy_(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            y.copy_(x)
        b.record()
        b.synchronize()
        return 2.0 * 4.0 * n * 5 / (a.elapsed_time(b) * 1e-3) / 1e9
    except Exception:   # noqa: BLE001
        return None


def kernel_source_hash():
    """Hash of the kernel sources the library is built from.  The counter passes are separate
    rocprofv3 runs (tools/profile_round.sh), so their summary in profiles/pmc_traffic.json carries
    the hash of the sources it was measured on; numbers from other sources are not reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fluidsolver_amd", "csrc")
    for name in sorted(os.listdir(d)):
        # the files of the kernels this benchmark launches; the host-side orchestration does not
        # change what a launch moves, and the optional stages' kernels (fs_mgpcg.hip, fs_vort.hip)
        # are not launched by it
        if name in ("fs_kernels.h", "fs_kernels.hip", "fs_jacobi.hip", "fs_advect.hip"):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def measure_traffic(n, iters, cfg, ran, flags):
    """HBM bytes one launch of the Jacobi kernel `ran` moves, measured now: two rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE cannot share one) over the headless C++ benchmark of the same library
    on the same workload, as child processes, with the kernel variant pinned to the one this run
    timed.  Returns (bytes per launch, description) or (None, reason).  FETCH_SIZE / WRITE_SIZE are
    in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced streams as 64 bytes,
    so the read side is doubled (tools/pmc_to_json.py applies the same correction)."""
    import csv
    import shutil
    import tempfile
    exe = os.path.join(ROOT, "fluidsolver_amd", "bin", "fs_bench")
    prof = shutil.which("rocprofv3")
    if not prof or not os.path.exists(exe):
        return None, "rocprofv3 or fluidsolver_amd/bin/fs_bench not found"
    env = dict(os.environ)
    deep = (cfg.get("sweeps_per_pass") or 4) > 4
    env.update(FS_JACOBI_TUNE="0", FS_JACOBI_CFG=str(max(cfg["variant"], 0)), FS_JACOBI_NT=str(cfg["nontemporal_stores"]),
               FS_JACOBI_DEEP="1" if deep else "0", TMPDIR="/tmp")
    if cfg["planes_per_chunk"] > 0:
        env["FS_JACOBI_LZ"] = str(cfg["planes_per_chunk"])
    if deep:
        env.update(FS_JACOBI_DEEP_LZ=str(cfg.get("deep_planes_per_chunk") or 0),
                   FS_JACOBI_DEEP_NT=str(cfg.get("deep_nontemporal_stores") or 0))
    got = {}
    work = tempfile.mkdtemp(prefix="fs_traffic_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            # (the program itself follows `--`: the profiler's library has initialised the GPU by
            # then, and a launcher in between would replace a process that has)
            cmd = ["timeout", "-s", "KILL", "120", prof, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p",
                   "--", exe, str(n), str(iters), "4", "2", "-", str(flags)]
            r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None, "the %s pass ended with status %d" % (counter, r.returncode)
            vals = []
            for dirpath, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        with open(os.path.join(dirpath, f)) as fh:
                            vals += [float(row["Counter_Value"]) for row in csv.DictReader(fh)
                                     if row["Counter_Name"] == counter and ran in row["Kernel_Name"]]
            if not vals:
                return None, "no %s rows for %s" % (counter, ran)
            got[counter] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    total = got["FETCH_SIZE"] * 1024.0 * 2.0 + got["WRITE_SIZE"] * 1024.0
    return total, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of fs_bench run by this invocation (reads x2)"


VALU_CYCLES_PER_INST = 2.0     # one wave-level vector instruction per SIMD every 2 cycles (nominal)
SIMDS = 1024                   # 256 CUs x 4


def load_pmc(kernel, workload, ran=None):
    """(entry, stale) of a kernel class from the committed rocprofv3 --pmc passes: HBM bytes per
    launch and wave-level VALU instructions per launch.  stale = measured on other kernel sources.
    `ran`: the kernel this run launched for the Jacobi passes (rocprofv3's name without the
    namespace); an entry measured on another variant (the timing at creation picked differently,
    or a switch forced one) is not this run's and is dropped."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e = d.get("kernels", {}).get(kernel, {}).get(workload)
        if e and kernel == "jacobi" and ran is not None:
            measured = (d.get("jacobi_kernel") or {}).get(workload, "")
            if ran not in measured:
                e = None
        return e, d.get("source_hash") != kernel_source_hash()
    except Exception:
        return None, True


# rocprofv3 kernel names of the stages (what `--kernel-trace --stats` lists)
def rocprof_kernel_name(stage, cfg, scheme, fused_div, slab):
    mac, halo = ("true" if scheme else "false"), ("true" if slab else "false")
    if stage == "jacobi":
        if cfg.get("sweeps_per_pass", 4) > 4:
            return "k_jacobi_deep<%d, 48, %d>" % (cfg["sweeps_per_pass"], cfg.get("deep_nontemporal_stores", 0))
        nty = cfg.get("tile_rows", 64)
        return "k_jacobi_tb<0, 4, %d, 1, %d, %d>" % (nty, cfg.get("aligned_rows", 0), cfg.get("nontemporal_stores", 0))
    if stage == "advect_velocity":
        return "k_advect_velocity_lds<DimsNarrow, %s, %s>" % (mac, halo)
    if stage == "advect_scalars":
        return "k_advect_scalars_lds<DimsNarrow, %s, %s, %s>" % (mac, halo, "true" if fused_div else "false")
    return {"gradient": "k_gradient4", "force": "k_force4", "divergence": "k_divergence",
            "source": "k_source"}.get(stage, stage)


def valu_roofline(pmc, ms_per_launch):
    """`roofline`-shaped entry of a kernel bound by vector-instruction issue: achieved = wave-level
    VALU instructions of one launch (committed counter pass) / its duration measured now; peak =
    1024 SIMDs x clock / 2 cycles per instruction, with the clock the chip held in the counter pass
    (GRBM_GUI_ACTIVE / 8 XCDs / duration of that pass)."""
    if not pmc or not pmc.get("SQ_INSTS_VALU") or ms_per_launch <= 0:
        return None
    clock = pmc.get("clock_ghz") or 2.4
    achieved = pmc["SQ_INSTS_VALU"] / (ms_per_launch * 1e-3) / 1e9
    peak = SIMDS * clock / VALU_CYCLES_PER_INST
    return {"bound": "valu", "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s",
            "frac": achieved / peak, "clock_ghz": clock,
            "clock_source": "GRBM_GUI_ACTIVE / 8 / duration of the committed counter pass" if pmc.get("clock_ghz")
                            else "nominal (no GRBM_GUI_ACTIVE in the committed counter pass)",
            "valu_insts_per_launch": pmc["SQ_INSTS_VALU"], "avg_launch_ms": ms_per_launch}


def stage_rooflines(prof, cells, steps, iters, n, jacobi_ran=None):
    """Per stage: algorithmic bytes per step / device time per step against the HBM peak, and for
    the gather kernels, whose limiter is instruction issue, the share of the chip's vector-issue
    rate (1024 SIMDs x one wave instruction per 2 cycles at the clock of the counter pass) their
    committed instruction count amounts to in the time measured now."""
    out = {}
    for name, (ms, launches) in prof.items():
        if ms <= 0 or name not in BYTES_PER_CELL:
            continue
        if name == "divergence" and ms / steps < 0.05:
            continue   # fused into the scalar advection: what is left are a slab's few outer planes
        bpc = BYTES_PER_CELL[name] * (iters if name == "jacobi" else 1)
        gbs = bpc * cells * steps / (ms * 1e-3) / 1e9
        e = {"ms_per_step": ms / steps, "algorithmic_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
        pmc, stale = load_pmc(name, "%d^3" % n, jacobi_ran)
        if pmc and not stale:
            per_step = launches / steps if name == "jacobi" else 1.0
            e["hbm_traffic_gbs"] = pmc["hbm_bytes_per_launch"] * per_step / (ms / steps * 1e-3) / 1e9
            e["hbm_traffic_frac"] = e["hbm_traffic_gbs"] / HBM_PEAK_GBS
        e["limiter"] = "hbm"
        if name.startswith("advect"):
            e["limiter"] = "valu issue"
            if pmc and not stale and pmc.get("SQ_INSTS_VALU"):
                # one launch per step: instructions per launch = per step
                v = valu_roofline(pmc, ms / steps)
                e["valu_issue_frac"] = v["frac"]
                e["valu_clock_ghz"] = v["clock_ghz"]
                e["valu_insts_per_cell"] = pmc["SQ_INSTS_VALU"] * 64.0 / cells
        out[name] = e
    return out


def launch_ranks(n, argv):
    """Run this script as n ranks under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1) and return the launcher's exit status.  Rank 0 prints the JSON line on the shared
    stdout."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("FS_BENCH_DRY_LAUNCH"):   # tests: show the command instead of running it
        print(json.dumps({"launch": cmd, "torch_imported": "torch" in sys.modules}))
        return 0
    return subprocess.call(cmd, env=env)


ABANDONED = []   # solvers whose transport failed: kept alive, never destroyed (see main)


# Transports of a z-slab run, tried in this order until one initialises and completes a step on
# every rank: RCCL with the production ordering; RCCL with the host, not the compute stream, waiting
# for received planes; RCCL with every transfer on the compute stream (no overlap); halos through
# host buffers and gloo.  (transport, environment switches read at solver creation, label)
TIERS = (
    ("rccl", {}, "RCCL send/recv on a communication stream"),
    ("rccl", {"FS_XCHG_HOST_WAIT": "1"}, "RCCL send/recv on a communication stream, host-side wait for received planes"),
    ("rccl", {"FS_XCHG_STREAMS": "0"}, "RCCL send/recv on the compute stream (no overlap)"),
    ("host", {}, "host buffers + gloo"),
)


def open_slab_solver(make, p, rank, world, want, warmup, all_min, environ=os.environ):
    """(solver, tier index, warm-up steps left) for one rank of a z-slab run.  `make(p, rank, world,
    transport=...)` builds a solver, `all_min(int)` is the minimum over the ranks.  Every rank must
    end up on the same tier, so each decision is taken on the minimum over all ranks of "it worked
    here".  A tier whose solver cannot be created, or whose first step (the first warm-up step,
    which doubles as the trial: a transfer that does not complete is reported by the library after
    FS_XCHG_TIMEOUT_MS instead of hanging) fails on any rank, is given up for the next one, so that
    the run still yields a number -- with RCCL as long as any ordering of it works."""
    first = 0 if want == "rccl" else len(TIERS) - 1
    for tier in range(first, len(TIERS)):
        transport, env, label = TIERS[tier]
        last = tier == len(TIERS) - 1
        saved = {k: environ.get(k) for k in env}
        environ.update(env)
        solver, ok = None, 1
        try:
            solver = make(p, rank, world, transport=transport).Initialize()
        except Exception as e:   # noqa: BLE001
            if last:
                raise
            print("bench.py rank %d: %s unavailable (%s)" % (rank, label, e), file=sys.stderr)
            ok = 0
        finally:
            for k, v in saved.items():
                if v is None:
                    environ.pop(k, None)
                else:
                    environ[k] = v
        if last:
            return solver, tier, warmup
        if all_min(ok) == 0:
            if solver is not None:
                solver.close()
            continue
        if warmup <= 0:
            return solver, tier, warmup
        try:
            solver.step(1)
            solver.sync()
        except Exception as e:   # noqa: BLE001
            print("bench.py rank %d: %s failed in the first step (%s)" % (rank, label, e), file=sys.stderr)
            ok = 0
        if all_min(ok) == 1:
            return solver, tier, warmup - 1
        # not closed: destroying a communicator with a transfer stuck in it may block
        ABANDONED.append(solver)
    raise AssertionError("unreachable")


def swirl_fields(n, seed=5):
    """A developed flow for timing the advection off its fast path: a smooth swirl with |u| dt of
    1-2 cells that is non-zero everywhere and crosses every wall, random density and temperature.
    Dense (z, y, x) float32 arrays keyed by field id."""
    import numpy as np
    amp = 6.25
    a = np.arange(n + 1, dtype=np.float64)
    f = 2.0 * np.pi / n
    sx, cx = np.sin(f * a)[None, None, :], np.cos(f * a)[None, None, :]
    sy, cy = np.sin(f * a)[None, :, None], np.cos(f * a)[None, :, None]
    sz, cz = np.sin(f * a)[:, None, None], np.cos(f * a)[:, None, None]
    rng = np.random.default_rng(seed)
    return {
        0: (amp * sy[:, :n] * cz[:n] + 0.3 * amp * cx).astype(np.float32),
        1: np.ascontiguousarray(np.broadcast_to((amp * sz[:n] * cx[:, :, :n] + 0.5).astype(np.float32), (n, n + 1, n))),
        2: np.ascontiguousarray(np.broadcast_to((amp * sx[:, :, :n] * cy[:, :n] - 0.25).astype(np.float32), (n + 1, n, n))),
        3: rng.random((n, n, n), dtype=np.float32),
        4: rng.standard_normal((n, n, n), dtype=np.float32),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="nx = ny = per-GPU nz")
    ap.add_argument("--iters", type=int, default=80, help="Jacobi sweeps per step")
    ap.add_argument("--scheme", type=int, default=1)
    ap.add_argument("--naive-jacobi", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=512,
                    help="edge of the cube the CPU oracle steps once for cpu_baseline (default: the workload itself)")
    ap.add_argument("--transport", choices=("rccl", "host"), default="rccl")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = size x size x (size*N) column, strong = size^3 cut into N slabs")
    ap.add_argument("--tolerance", type=float, default=0.0,
                    help="> 0: convergence mode, sweep in groups of --check-every until sum r^2 (all ranks) "
                         "is below it or --iters sweeps are done")
    ap.add_argument("--check-every", type=int, default=8)
    ap.add_argument("--fmad", action="store_true",
                    help="FS_FLAG_FMAD: the contracted arithmetic of the advection and the buoyancy (not the default contract)")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="N = 1: measure roofline.traffic now with two rocprofv3 --pmc passes of fluidsolver_amd/bin/fs_bench "
                         "(child processes, ~10 s) instead of reading the committed profiles/pmc_traffic.json")
    ap.add_argument("--flow", choices=("plume", "swirl"), default="plume",
                    help="plume = BASELINE's workload (zero state + sphere source); swirl = a developed flow uploaded "
                         "before the warm-up steps (times the advection off its all-lanes-in-brick fast path; N = 1)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become the launcher.  Nothing in this
        # process has touched torch or the GPU; the ranks are fresh children.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # this driver shares device memory between processes through dmabuf only; without the setting
    # RCCL fails with hipIpcGetMemHandle: invalid argument (read when the HSA runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # gloo and RCCL announce themselves on stdout (connections, a version banner); the one JSON
    # line must stay alone there, so everything else this process and its libraries print goes to
    # stderr and the line is written to the original stdout at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import fluidsolver_amd as fs

    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # one GPU per rank on the node; on a box with fewer GPUs than ranks (rehearsals) ranks share
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)

    dist = None
    if world > 1:
        # torch.distributed is the control plane only (unique-id hand-out, barriers, max over
        # ranks); the halos travel inside the library over RCCL
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n = args.size
    nz_total = n * world if args.scaling == "weak" else n
    flags = (fs.FS_FLAG_JACOBI_NAIVE if args.naive_jacobi else 0) | (fs.FS_FLAG_FMAD if args.fmad else 0)
    conv = {}
    if args.tolerance > 0:
        conv = dict(tolerance=args.tolerance, check_every=args.check_every, max_sweeps=args.iters)
    warm_left = args.warmup
    if world == 1:
        p = fs.default_params(n, n, n, jacobi_iters=args.iters, scheme=args.scheme,
                              device=device, flags=flags, **conv)
        solver = fs.Solver(p).Initialize()
        cells_total = n ** 3
        workload = "%d^3 smoke plume, %d Jacobi sweeps/step, MacCormack, 1 GPU" % (n, args.iters)
        if args.flow == "swirl":
            for fid, arr in swirl_fields(n).items():
                solver.upload(fid, arr)
            workload = ("%d^3 developed swirl (|u| dt up to 1.6 cells, non-zero everywhere) + sphere source, "
                        "%d Jacobi sweeps/step, MacCormack, 1 GPU" % (n, args.iters))
        if args.fmad:
            workload += ", fused multiply-adds (FS_FLAG_FMAD)"
        par = "1 GPU"
        tier = None
    else:
        from fluidsolver_amd.slab import SlabSolver
        p = fs.default_params(n, n, nz_total, jacobi_iters=args.iters, scheme=args.scheme,
                              device=device, flags=flags, **conv)

        def all_min(v):
            flag = torch.tensor([v])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item())

        if args.flow != "plume":
            raise SystemExit("bench.py: --flow swirl is a single-GPU workload")
        # a transfer that never completes must end its tier well before the driver's patience does
        os.environ.setdefault("FS_XCHG_TIMEOUT_MS", "20000")
        solver, tier, warm_left = open_slab_solver(SlabSolver, p, rank, world, args.transport, args.warmup, all_min)
        cells_total = n * n * nz_total
        workload = ("%dx%dx%d %s, %d Jacobi sweeps/step, MacCormack, z-slabs on %d GPUs, "
                    "halos over %s" % (n, n, nz_total,
                                       "weak-scaling column" if args.scaling == "weak" else "strong scaling",
                                       args.iters, world, TIERS[tier][2]))
        par = "z-slab x%d" % world

    def barrier():
        solver.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    solver.step(warm_left)
    barrier()
    solver.profile(True)
    solver.profile_reset()
    sweeps_done = args.iters * args.steps
    t0 = time.perf_counter()
    if args.tolerance > 0:
        # the sweep count of a step is decided by the residual: count what actually ran
        sweeps_done = 0
        for _ in range(args.steps):
            solver.step(1)
            sweeps_done += solver.last_sweeps
    else:
        solver.step(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = solver.profile_read()
    halo = solver.halo_wait()
    solver.profile(False)

    if dist is not None:
        # max over the ranks of the wall time and of the time each rank waited for halos
        t = torch.tensor([elapsed, halo["host_ms"], halo["device_ms"]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, halo["host_ms"], halo["device_ms"] = (float(v) for v in t)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = cells_total * args.steps / elapsed
        cells_rank = cells_total // world
        # `roofline` is for the HBM-bound kernel class with the largest device time (the Jacobi
        # passes whenever sweeps run); the gather kernels, which take a similar share of the step, are
        # bound by vector-instruction issue and priced against that in `roofline_secondary`
        largest = max(prof, key=lambda k: prof[k][0])
        hbm_bound = {k: v for k, v in prof.items() if k in BYTES_PER_CELL and not k.startswith("advect") and v[0] > 0}
        dom = max(hbm_bound, key=lambda k: hbm_bound[k][0]) if hbm_bound else largest
        dom_ms, dom_launches = prof[dom]
        sweeps_per_launch = 1.0
        if dom == "jacobi":
            sweeps_per_launch = sweeps_done / max(dom_launches, 1)
            # what a pass must move once: read p, read div, write p'
            algo_bytes = BYTES_PER_CELL["jacobi"] * cells_rank
        else:
            # advect_velocity makes 3 launches per step; its bytes are per stage
            algo_bytes = BYTES_PER_CELL.get(dom, 0.0) * cells_rank * args.steps / max(dom_launches, 1)
        avg_launch_s = dom_ms / max(dom_launches, 1) * 1e-3
        achieved = algo_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        cfg = solver.jacobi_config()
        fused_div = world == 1
        jacobi_ran = rocprof_kernel_name("jacobi", cfg, args.scheme, fused_div, world > 1)
        pmc, stale = load_pmc(dom, "%d^3" % n, jacobi_ran)
        usable = pmc and not stale and world == 1 and args.flow == "plume" and not args.fmad
        traffic = float(pmc["hbm_bytes_per_launch"]) if usable else None
        traffic_source = None if traffic is None else "profiles/pmc_traffic.json (%s)" % pmc.get("source", "?")
        if args.measure_traffic and world == 1 and dom == "jacobi" and args.flow == "plume":
            live, how = measure_traffic(n, args.iters, cfg, jacobi_ran, flags)
            if live is not None:
                traffic, traffic_source = live, how
            else:
                print("bench.py: --measure-traffic: %s; keeping %s" % (how, traffic_source), file=sys.stderr)
        stage_ms = {k: round(v[0] / args.steps, 4) for k, v in prof.items()}
        out = {
            "metric": "cell-updates/sec per step (incl. pressure iters)",
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else None,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (zero initial state, sphere source every step, no RNG)" if args.flow == "plume" else
                    "synthetic (analytic swirl velocity, seeded random density and temperature, sphere source every step)",
            "config": {"workload": workload, "nx": n, "ny": n, "nz": nz_total,
                       "jacobi_sweeps": args.iters, "parallelism": par, "flow": args.flow},
            # B_step = 144 + 12 K bytes per cell, every sweep credited: > 1 is possible through temporal blocking
            "step_hbm_frac_credited": step_bytes_per_cell(sweeps_done / args.steps) * cells_rank / (ms_per_step * 1e-3)
                                      / 1e9 / HBM_PEAK_GBS,
            "stage_ms_per_step": stage_ms,
            # secondary metric: lattice updates per second of the pressure sweeps alone (per GPU)
            "jacobi_lups": (cells_rank * sweeps_done / (prof["jacobi"][0] * 1e-3)) if prof.get("jacobi", (0, 0))[0] > 0 else None,
            "stage_roofline": stage_rooflines(prof, cells_rank, args.steps, sweeps_done / args.steps, n, jacobi_ran),
            "autotune": {"jacobi_tile_variant": cfg["variant"], "jacobi_tile_rows": cfg["tile_rows"],
                         "jacobi_aligned_rows": cfg["aligned_rows"], "jacobi_planes_per_chunk": cfg["planes_per_chunk"],
                         "jacobi_nontemporal_stores": cfg["nontemporal_stores"],
                         "jacobi_sweeps_per_pass": cfg.get("sweeps_per_pass"),
                         "jacobi_deep_planes_per_chunk": cfg.get("deep_planes_per_chunk"),
                         "jacobi_deep_nontemporal_stores": cfg.get("deep_nontemporal_stores")},
            "rccl_ranks": solver.comm_ranks if world > 1 else 0,
            "largest_stage": largest,
            "roofline": {
                "stage": dom,
                "kernel": rocprof_kernel_name(dom, cfg, args.scheme, fused_div, world > 1),
                "bound": "hbm",
                # what holds this kernel back in practice (the gather kernels are bound by vector
                # instruction issue, not by memory: see roofline_secondary)
                "limiter": "valu issue" if dom.startswith("advect") else "hbm",
                # bytes one launch must move once (a fused pass of several sweeps: read p, read div,
                # write p' = 12 B per cell, however many sweeps it fuses) / its average duration
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # every fused sweep credited its own 12 B per cell (the algorithmic figure of a sweep
                # times the sweeps of the launch): exceeds the peak by construction of the blocking
                "sweeps_per_launch": sweeps_per_launch,
                "achieved_credited": achieved * sweeps_per_launch,
                "frac_credited": achieved * sweeps_per_launch / HBM_PEAK_GBS,
                # the bytes the fabric actually moved per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE
                # of the same kernel sources, from the committed counter pass)
                "traffic": traffic,
                "frac_real": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic and avg_launch_s > 0 else None,
                "traffic_source": traffic_source,
                "avg_launch_ms": avg_launch_s * 1e3,
                # what to compare between kernel variants: device time per sweep and cell, not `frac`
                # (a pass that fuses more sweeps moves the same bytes in a longer launch, so its
                # `frac` and `frac_real` fall while its sweeps per second rise)
                "us_per_sweep": (avg_launch_s * 1e6 / sweeps_per_launch) if sweeps_per_launch > 0 else None,
                "launches": dom_launches,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        # the gather kernel: bound by vector-instruction issue, priced against that
        av_ms, av_launches = prof.get("advect_velocity", (0.0, 0))
        apmc, astale = load_pmc("advect_velocity", "%d^3" % n)
        sec = valu_roofline(apmc if (apmc and not astale and world == 1 and args.flow == "plume" and not args.fmad) else None,
                            av_ms / args.steps)
        if sec:
            sec["stage"] = "advect_velocity"
            sec["kernel"] = rocprof_kernel_name("advect_velocity", cfg, args.scheme, fused_div, world > 1)
            out["roofline_secondary"] = sec
        if world > 1:
            out["transport_tier"] = tier
            out["halo_wait_ms_per_step"] = {"host": halo["host_ms"] / args.steps, "device": halo["device_ms"] / args.steps,
                                            "note": "max over ranks; host = host time inside blocking exchange calls, "
                                                    "device = compute stream standing still for received planes"}
            out["exchanges_per_step"] = halo["exchanges"] / args.steps
        if traffic is None and world == 1:
            out["roofline"]["traffic_note"] = ("no counter pass of these kernel sources is committed"
                                               if stale else "no counter entry for this workload and kernel variant")
        if args.tolerance > 0:
            out["config"]["tolerance"] = args.tolerance
            out["config"]["check_every"] = args.check_every
            out["config"]["sweeps_per_step_run"] = sweeps_done / args.steps
        if world == 1:
            out["roofline"]["measured_copy_gbs"] = measured_copy_bandwidth(torch, device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.iters, args.cpu_sample)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    solver.close()
    if dist is not None:
        dist.destroy_process_group()
    if ABANDONED:
        # leave without running the destructors of a solver whose transfers never completed
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
