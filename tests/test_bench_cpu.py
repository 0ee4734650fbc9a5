"""bench.py host logic that needs no GPU: `python bench.py --gpus N` started without a launcher
becomes the launcher of N fresh ranks (before anything touches torch or the GPU), and the counter
summary is only reported for the kernel sources it was measured on."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)


def test_plain_invocation_with_several_gpus_takes_the_launcher_path():
    r = run_bench(["--gpus", "4", "--steps", "3", "--warmup", "1"], {"FS_BENCH_DRY_LAUNCH": "1"})
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["torch_imported"] is False
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")


def test_rank_count_mismatch_is_an_error():
    r = run_bench(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_counter_summary_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"hbm_bytes_per_launch": 123.0, "SQ_INSTS_VALU": 5.0}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    (prof / "pmc_traffic.json").write_text(json.dumps({"source_hash": h, "kernels": {"jacobi": {"512^3": entry}}}))
    assert bench.load_pmc("jacobi", "512^3") == (entry, False)
    assert bench.load_pmc("jacobi", "256^3") == (None, False)
    (prof / "pmc_traffic.json").write_text(json.dumps({"source_hash": "other", "kernels": {"jacobi": {"512^3": entry}}}))
    assert bench.load_pmc("jacobi", "512^3") == (entry, True)
    # ... and to the variant of the Jacobi kernel the counters were collected on
    ran = "k_jacobi_deep<6, 48, 1>"
    (prof / "pmc_traffic.json").write_text(json.dumps({
        "source_hash": h, "jacobi_kernel": {"512^3": "void (anonymous namespace)::%s((anonymous namespace)::TbArgs)" % ran},
        "kernels": {"jacobi": {"512^3": entry}, "advect_velocity": {"512^3": entry}}}))
    assert bench.load_pmc("jacobi", "512^3", ran) == (entry, False)
    assert bench.load_pmc("jacobi", "512^3", "k_jacobi_tb<0, 4, 64, 1, 1, 1>") == (None, False)
    assert bench.load_pmc("advect_velocity", "512^3", "k_jacobi_tb<0, 4, 64, 1, 1, 1>") == (entry, False)


def test_live_traffic_measurement_reports_why_it_cannot_run(monkeypatch):
    """--measure-traffic without a profiler on the PATH: no number, a reason, and the caller keeps
    the committed summary."""
    sys.path.insert(0, ROOT)
    import bench
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    cfg = {"variant": 2, "nontemporal_stores": 1, "planes_per_chunk": 103, "sweeps_per_pass": 6,
           "deep_planes_per_chunk": 171, "deep_nontemporal_stores": 1}
    got, why = bench.measure_traffic(512, 80, cfg, "k_jacobi_deep<6, 48, 1>", 0)
    assert got is None and "rocprofv3" in why


ENV = {}     # the environment open_slab_solver edits (a stand-in for os.environ)


def _tier_of(transport):
    if transport == "host":
        return 3
    return 1 if ENV.get("FS_XCHG_HOST_WAIT") == "1" else 2 if ENV.get("FS_XCHG_STREAMS") == "0" else 0


class _FakeSolver:
    """Stand-in for SlabSolver: `plan` says which tiers fail to initialise or to step.  Like the
    library, it reads the environment switches when it is created."""
    plan = {}
    made = []

    def __init__(self, p, rank, world, transport):
        self.transport, self.steps, self.closed = transport, 0, False
        self.tier = None
        _FakeSolver.made.append(self)

    def Initialize(self):
        self.tier = _tier_of(self.transport)
        if _FakeSolver.plan.get(self.tier) == "init":
            raise RuntimeError("no such transport here")
        return self

    def step(self, n=1):
        if _FakeSolver.plan.get(self.tier) == "step":
            raise RuntimeError("the transfer did not finish")
        self.steps += n

    def sync(self):
        pass

    def close(self):
        self.closed = True


@pytest.mark.parametrize("plan,want,tier,warm_left,abandoned", [
    ({}, "rccl", 0, 2, 0),                               # the trial step is the first warm-up step
    ({0: "init", 1: "init", 2: "init"}, "rccl", 3, 3, 0),   # RCCL cannot be set up at all: host-staged halos
    ({0: "step"}, "rccl", 1, 2, 1),                      # the GPU-side wait stalls: RCCL with the host waiting
    ({0: "step", 1: "step"}, "rccl", 2, 2, 2),           # ... then RCCL on the compute stream alone
    ({0: "step", 1: "step", 2: "step"}, "rccl", 3, 3, 3),   # ... and only then host buffers + gloo
    ({}, "host", 3, 3, 0),
])
def test_bench_transport_selection(plan, want, tier, warm_left, abandoned, monkeypatch):
    """The tiers of a multi-rank run: three orderings of the RCCL transport before the host-staged
    fallback; the switches of a tier are set while its solver is created and restored afterwards;
    a solver whose transfer stalled is abandoned, never closed."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ABANDONED", [])
    ENV.clear()
    ENV["FS_XCHG_STREAMS"] = "keep"       # a value the caller had set comes back after a tier that overrides it
    _FakeSolver.plan, _FakeSolver.made = plan, []
    s, t, w = bench.open_slab_solver(_FakeSolver, None, 0, 2, want, 3, lambda v: v, environ=ENV)
    assert (t, w, len(bench.ABANDONED)) == (tier, warm_left, abandoned)
    assert s.tier == tier and not s.closed and s.transport == bench.TIERS[tier][0]
    assert ENV == {"FS_XCHG_STREAMS": "keep"}
    for other in _FakeSolver.made:
        assert other is s or not other.closed      # failed to initialise, or abandoned: never closed
    # RCCL fine here but not on another rank: this rank's RCCL solvers are closed, all use the host transport
    ENV.clear()
    _FakeSolver.plan, _FakeSolver.made = {}, []
    s, t, w = bench.open_slab_solver(_FakeSolver, None, 0, 2, "rccl", 3, lambda v: 0, environ=ENV)
    if want == "rccl":
        assert t == 3 and all(m.closed for m in _FakeSolver.made[:3]) and s is _FakeSolver.made[3]
    # without warm-up steps there is no trial
    _FakeSolver.plan, _FakeSolver.made = {0: "step"}, []
    s, t, w = bench.open_slab_solver(_FakeSolver, None, 0, 2, "rccl", 0, lambda v: v, environ=ENV)
    assert (t, w) == (0, 0)


def test_swirl_flow_is_developed_everywhere():
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    f = bench.swirl_fields(24)
    assert f[0].shape == (24, 24, 25) and f[1].shape == (24, 25, 24) and f[2].shape == (25, 24, 24)
    speed = np.sqrt(f[0][:, :, :24] ** 2 + f[1][:, :24, :] ** 2 + f[2][:24] ** 2)
    assert 0.2 * speed.max() <= 8.0 and 1.0 <= 0.2 * speed.max()        # |u| dt of 1-2 cells at its fastest
    assert (speed > 0.05).mean() > 0.99
