"""Multi-rank z-slab runs against the single-domain oracle (bit-exact).

The ranks share the one GPU of the test box and move their halos through host buffers and
gloo (`transport="host"`): everything except the RCCL calls themselves is the production
path (slab kernels, ghost-plane bookkeeping, fused Jacobi passes with wide halos, residual
all-reduce)."""
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
    env = dict(os.environ, OMP_NUM_THREADS="4", **(env_extra or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=280, text=True)
    assert r.returncode == 0 and "SLAB_OK" in r.stdout, r.stdout[-3000:]


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
    run_ranks(1, ["selfpeer", 64, 48, 24, 12, 20], 29661)


def test_single_rank_rccl_communicator():
    """world = 1 through the RCCL code path: loads librccl, creates the unique id."""
    run_ranks(1, ["rccl", 16, 16, 16, 2, 8], 29641)

