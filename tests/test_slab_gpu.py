"""Multi-rank z-slab runs against the single-domain oracle (bit-exact).

The ranks share the one GPU of the test box and move their halos through host buffers and
gloo (`transport="host"`): everything except the RCCL calls themselves is the production
path (slab kernels, ghost-plane bookkeeping, fused Jacobi passes with wide halos, transfers on
the communication stream ordered by two events per exchange, residual all-reduce)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, args, port, env_extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "slab_worker.py")] + [str(a) for a in args]
    # a stuck worker dumps its Python stacks and exits after 100 s, a stalled exchange reports the
    # queue states after 40 s: either ends the run well before the launcher's own limit
    env = dict({"FS_WORKER_DUMP_AFTER": "100", "FS_XCHG_TIMEOUT_MS": "40000"}, **os.environ)
    # (every rank steps its own oracle: small teams that sleep in their barriers, or the ranks'
    #  spinning threads starve one another on a box with few cores)
    env.update(OMP_NUM_THREADS="4", OMP_WAIT_POLICY="passive", **(env_extra or {}))
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=160, text=True)
    except subprocess.TimeoutExpired as e:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "slab_fail_%d.log" % port), "w") as f:
            f.write((e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""))
        raise
    if r.returncode != 0 or "SLAB_OK" not in r.stdout:
        # keep the whole output of a failed run (the library's stall report precedes the
        # launcher's traceback) where a GPU-box run returns it
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "slab_fail_%d.log" % port), "w") as f:
            f.write(r.stdout)
        lines = [ln for ln in r.stdout.splitlines() if "[fluidsolver" in ln or "differs" in ln or "SLAB_" in ln]
        raise AssertionError("\n".join(lines[-40:]) + "\n...\n" + r.stdout[-1500:])


@pytest.mark.parametrize("world,dims,steps,k,extra", [
    (2, (32, 24, 32), 6, 20, []),            # fused passes, wide p halos
    (3, (24, 20, 31), 5, 7, []),             # ragged slabs (11,10,10), K not a multiple of 4
    (2, (18, 16, 24), 4, 6, ["naive"]),      # nx % 4 != 0 -> single sweeps, one plane per sweep
    (2, (32, 16, 26), 3, 8, ["random"]),     # random initial state uploaded per rank
    (4, (16, 16, 40), 4, 12, []),            # interior ranks have two neighbours
    (3, (20, 16, 48), 3, 8, ["random"]),     # 16 planes per rank: the interior rank also advects its
                                             # inner planes while the ghost planes travel
])
def test_slab_matches_oracle(world, dims, steps, k, extra):
    run_ranks(world, ["host", *dims, steps, k, *extra], 29611 + world)


def test_slab_tall_tiles_multi_tile():
    """64-row Jacobi tiles (the large-grid variant) with several tiles per plane, 7 sweeps: the slab
    passes of 4 and 3 sweeps read 4 / 3 ghost planes and recompute the intermediate ones."""
    run_ranks(2, ["host", 72, 68, 24, 3, 7], 29621, {"FS_JACOBI_CFG": "0"})


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("world,k", [(2, 7), (3, 8)])
def test_slab_every_tile_variant(cfg, world, k):
    """Every tile variant of the blocked Jacobi kernel (64 / 32 rows, plain / 64-byte aligned output
    rows) on slabs, several tiles per plane in x and y (132 x 70), passes of 4 and 3 sweeps; the
    3-rank run has an interior rank whose boundary launch covers two plane ranges.  On a slab the
    variant is otherwise whatever the timing at creation picks, so it is pinned here."""
    run_ranks(world, ["host", 132, 70, 40, 2, k, "random"], 29670 + 4 * world + cfg,
              {"FS_JACOBI_CFG": str(cfg), "FS_JACOBI_TUNE": "0"})


@pytest.mark.parametrize("cfg", [0, 3])
def test_slab_boundary_launch_with_several_chunks_per_range(cfg):
    """FS_JACOBI_LZ=2: the 4 boundary planes next to each neighbour are cut into two z chunks, so
    the interior rank's boundary launch has two chunks in each of its two output ranges."""
    run_ranks(3, ["host", 68, 40, 36, 2, 8, "random"], 29690 + cfg,
              {"FS_JACOBI_CFG": str(cfg), "FS_JACOBI_LZ": "2", "FS_JACOBI_TUNE": "0"})


@pytest.mark.parametrize("pghost", [8, 4])
@pytest.mark.parametrize("world,dims,steps,k", [
    (2, (36, 28, 56), 3, 20),     # 28 planes per rank: two pairs of passes and a single pass per step
    (3, (68, 40, 81), 2, 19),     # interior rank with two neighbours; pairs, then 3 sweeps in one pass
])
def test_slab_paired_passes_share_one_exchange(world, dims, steps, k, pghost):
    """Slabs of at least 25 planes hold 8 ghost planes of the pressure and run the fused Jacobi
    passes in pairs with one exchange per pair (pass A also produces 4 planes of each neighbour's
    slab); FS_SLAB_PGHOST=4 keeps one exchange per pass on the same grids."""
    run_ranks(world, ["host", *dims, steps, k, "random"], 29720 + world + pghost,
              {"FS_SLAB_PGHOST": str(pghost), "FS_JACOBI_TUNE": "0"})


@pytest.mark.parametrize("world,dims,steps,k,env", [
    # 28 planes per rank, 20 sweeps: two six-sweep passes (the second sends 8 planes), one pair of four-sweep passes
    (2, (72, 52, 56), 2, 20, {}),
    # interior rank with two neighbours; 26 sweeps: three six-sweep passes and a pair
    (3, (68, 40, 81), 2, 26, {"FS_JACOBI_TUNE": "0"}),
    # 19 planes per rank: too thin for pairs; three six-sweep passes and one single sweep, each with its exchange
    (3, (36, 28, 57), 2, 19, {}),
    # the sweep count of the benchmark: twelve six-sweep passes and one pair; chunks of 5 planes
    (2, (64, 100, 64), 1, 80, {"FS_JACOBI_DEEP_LZ": "5", "FS_JACOBI_DEEP_NT": "1"}),
    # four ranks; 12 sweeps are two six-sweep passes and nothing else
    (4, (48, 40, 80), 3, 12, {}),
])
def test_slab_six_sweep_passes(world, dims, steps, k, env):
    """FS_JACOBI_DEEP=1 on slabs of at least 17 planes with 8 ghost planes: passes of six sweeps from
    six valid ghost planes, the boundary planes next to each neighbour first, one exchange per pass
    hidden behind the interior; the last six-sweep pass sends the 8 planes the pair of four-sweep
    passes after it wants."""
    run_ranks(world, ["host", *dims, steps, k, "random", "deep"], 29771 + 2 * world + k % 11, dict(env, FS_JACOBI_DEEP="1"))


def test_slab_six_sweep_passes_need_thick_slabs_and_deep_ghosts():
    """Slabs of 16 planes, or ghost planes 4 deep, keep the four-sweep passes whatever the switch says
    (and still match the oracle)."""
    run_ranks(2, ["host", 32, 24, 32, 2, 12, "random", "four"], 29768, {"FS_JACOBI_DEEP": "1"})
    run_ranks(2, ["host", 36, 28, 56, 2, 12, "random", "four"], 29769, {"FS_JACOBI_DEEP": "1", "FS_SLAB_PGHOST": "4"})


@pytest.mark.parametrize("lattice", [1, 0])
@pytest.mark.parametrize("world,dims,steps,k,extra", [
    (2, (24, 20, 32), 4, 8, ["confine"]),
    (3, (20, 16, 48), 3, 8, ["confine", "random"]),   # 8 halo planes of vorticity, 16 planes per rank
])
def test_slab_vorticity_confinement(world, dims, steps, k, extra, lattice):
    """FS_FLAG_VORT_CONFINE on slabs: every rank holds its window of the vorticity arrays' flat
    index space and receives the planes its gathers reach in the neighbours' (the mismatched
    strides of the reference land a gather several planes above its face)."""
    run_ranks(world, ["host", *dims, steps, k, *extra], 29740 + world + 3 * lattice, {"FS_VORT_LATTICE": str(lattice)})


def test_slab_confinement_needs_slabs_as_thick_as_its_reach(fs):
    """A decomposition whose slabs are thinner than the reach of the confinement gathers is
    refused at creation."""
    import ctypes as C
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)
    p = fs.default_params(16, 16, 40, flags=fs.FS_FLAG_VORT_CONFINE)   # reach 8 planes, 5 per rank
    h = C.c_void_p()
    cb = slabmod._EXCHANGE_FN(lambda *a: 0)
    assert L.fs_create_slab(C.byref(p), 0, 8, None, cb, None, C.byref(h)) == -1
    assert b"upper neighbour" in L.fs_last_error()


@pytest.mark.parametrize("world,dims,steps,extra,env", [
    (2, (16, 24, 32), 4, ["mgpcg"], {}),                        # 16 planes per rank, coarsest level 2 planes
    (3, (16, 16, 96), 3, ["mgpcg", "random"], {}),              # 32 planes per rank, interior rank
    (2, (24, 16, 64), 3, ["mgpcg", "confine"], {}),             # the configuration the reference ships
    # every level the blocked kernel can take goes through it, four colours per pass (ghost depth 4)
    (2, (32, 32, 64), 3, ["mgpcg", "random"], {"FS_MG_TB_MIN": "0", "FS_MG_SLAB_GHOST": "4"}),
    # ... two colours per pass (ghost depth 2), three ranks
    (3, (32, 16, 96), 2, ["mgpcg"], {"FS_MG_TB_MIN": "0", "FS_MG_SLAB_GHOST": "2"}),
    # one ghost plane: an exchange before every colour
    (2, (16, 16, 32), 2, ["mgpcg"], {"FS_MG_SLAB_GHOST": "1"}),
    # coarsest slabs of odd thickness (9 and 5 planes) on the blocked path: the upper ranks' windows
    # start on planes of odd global index, the colour of a cell must follow the global index and the
    # sequence of exchanges must not depend on it
    (2, (32, 32, 36), 2, ["mgpcg", "random", "mglevel=2"], {"FS_MG_TB_MIN": "0", "FS_MG_SLAB_GHOST": "4"}),
    (3, (32, 32, 30), 2, ["mgpcg", "mglevel=2"], {"FS_MG_TB_MIN": "0", "FS_MG_SLAB_GHOST": "3"}),
])
def test_slab_mgpcg(world, dims, steps, extra, env):
    """FS_PRESSURE_MGPCG on slabs: every rank runs the V-cycle and the CG vector operations on its
    window of each level (owned planes plus ghost planes that are refreshed before the error of the
    window's edge can reach an owned plane), dot products are all-reduced.  Iteration counts equal
    the single-domain oracle's, fields agree to 1e-5."""
    port = 29750 + world + 3 * len(extra) + 7 * len(env) + (11 if "FS_MG_SLAB_GHOST" in env and env["FS_MG_SLAB_GHOST"] == "2" else 0)
    run_ranks(world, ["host", *dims, steps, 30, *extra], port, env)


def test_slab_mgpcg_needs_slabs_that_coarsen_on_their_own(fs):
    import ctypes as C
    from fluidsolver_amd import slab as slabmod
    L = fs.load_library()
    slabmod._bind(L)
    cb = slabmod._EXCHANGE_FN(lambda *a: 0)
    h = C.c_void_p()
    p = fs.default_params(16, 16, 40, pressure_solver=fs.FS_PRESSURE_MGPCG)     # 40 = 5 x 8, but 2 ranks need 16 | nz
    assert L.fs_create_slab(C.byref(p), 0, 2, None, cb, None, C.byref(h)) == -1
    assert b"multiple of 16" in L.fs_last_error()


def test_slab_one_stream_ordering_fallback():
    """FS_XCHG_STREAMS=0: the fall-back ordering with every transfer on the compute stream (no
    overlap) gives the same fields."""
    run_ranks(3, ["host", 20, 16, 48, 3, 8, "random"], 29698, {"FS_XCHG_STREAMS": "0"})


def test_slab_two_stream_ordering_with_tracing():
    """Four ranks on one GPU with the transfers on the communication stream (the default ordering
    of both transports; on this transport the host, not the compute stream, waits for the received
    planes) and FS_DEBUG_XCHG tracing on: every wait inside an exchange is bounded and a stall
    would end the run with a report of the operation at the head of each queue."""
    run_ranks(4, ["host", 16, 16, 40, 4, 12], 29699,
              {"FS_XCHG_STREAMS": "1", "FS_DEBUG_XCHG": "1", "FS_XCHG_TIMEOUT_MS": "30000",
               "FS_WORKER_DUMP_AFTER": "120"})


def test_slab_convergence_mode():
    """tolerance > 0: the residual norm is all-reduced over the ranks every check_every sweeps."""
    run_ranks(2, ["host", 24, 24, 24, 3, 0, "converge"], 29631)


def test_slab_reports_a_back_trace_beyond_the_ghost_planes():
    """|uz| dt = 6 planes against 4 ghost planes: FS_ERR_HALO on every rank with a lower
    neighbour (also from the inner planes, which are advected before the ghost planes arrive)."""
    run_ranks(3, ["host", 16, 16, 48, 1, 4, "toofast"], 29651)


def test_rccl_send_recv_and_stream_ordering_with_itself_as_neighbour():
    """RCCL on one GPU: the middle slab of three exchanges with itself (see slab_worker.self_peer);
    two-stream ordering against one-stream ordering, 12 steps of 7 exchanges each."""
    run_ranks(1, ["selfpeer", 64, 48, 24, 12, 20], 29661, {"FS_DEBUG_XCHG": "0"})   # tracing would keep the steps eager


def test_rccl_six_sweep_passes_with_itself_as_neighbour():
    """The same with six-sweep passes (24 planes, 8 ghost planes: two six-sweep passes and two
    four-sweep passes per step, each with its exchange)."""
    run_ranks(1, ["selfpeer", 64, 48, 24, 6, 20], 29664, {"FS_DEBUG_XCHG": "0", "FS_JACOBI_DEEP": "1"})


def test_rccl_mgpcg_with_itself_as_neighbour():
    """MGPCG + vorticity confinement on the RCCL transport (window halos over send/recv, dot products
    over ncclAllReduce) against the host-staged transport, see slab_worker.self_peer_mgpcg."""
    run_ranks(1, ["selfpeer_mgpcg", 32, 24, 24, 3, 30], 29663)


def test_host_transport_stream_ordering_with_itself_as_neighbour():
    """The host-staged transport in ONE process (the exchange function loops the planes back) with
    the compute stream itself waiting for the received planes (FS_XCHG_HOST_WAIT=0, the ordering of
    the RCCL transport): two-stream ordering against one-stream ordering, 12 steps."""
    run_ranks(1, ["selfpeer_host", 64, 48, 24, 12, 20], 29662, {"FS_XCHG_HOST_WAIT": "0"})


def _devices():
    import torch
    return torch.cuda.device_count()     # (counting devices does not initialise the GPU)


RCCL_CASES = [
    (2, (64, 48, 64), 4, 20, ["random"], {}),                       # 32 planes per rank: paired passes, advection split
    (2, (36, 28, 24), 3, 7, [], {"FS_SLAB_PGHOST": "4"}),            # one exchange per pass, passes of 4 and 3 sweeps
    (2, (32, 24, 32), 3, 30, ["mgpcg", "confine"], {}),              # the shipped configuration: windows + all-reduce
    (2, (24, 24, 24), 3, 0, ["converge"], {}),                       # residual norm over ncclAllReduce
    (2, (64, 48, 64), 3, 20, ["random"], {"FS_XCHG_HOST_WAIT": "1"}),   # the fall-back orderings of the same transport
    (2, (64, 48, 64), 3, 20, ["random"], {"FS_XCHG_STREAMS": "0"}),
    (4, (48, 40, 80), 3, 12, ["random"], {}),                        # interior ranks with two neighbours
    (8, (64, 64, 256), 2, 16, [], {}),                               # the strong-scaling layout of an 8-GPU node
    (2, (64, 48, 64), 3, 20, ["random", "deep"], {"FS_JACOBI_DEEP": "1"}),   # six-sweep passes, then a pair
]


@pytest.mark.parametrize("case", range(len(RCCL_CASES)))
def test_rccl_between_devices(case):
    """One rank per GPU, halos over ncclSend/ncclRecv between DISTINCT devices (and the residual
    norm / MGPCG's dot products over ncclAllReduce), bit for bit against the single-domain oracle.
    Needs as many GPUs as ranks: skipped on a smaller box."""
    world, dims, steps, k, extra, env = RCCL_CASES[case]
    if _devices() < world:
        pytest.skip("%d GPUs needed, %d visible" % (world, _devices()))
    run_ranks(world, ["rccl", *dims, steps, k, *extra], 29800 + case, env)


def test_single_rank_rccl_communicator():
    """world = 1 through the RCCL code path: loads librccl, creates the unique id."""
    run_ranks(1, ["rccl", 16, 16, 16, 2, 8], 29641)

