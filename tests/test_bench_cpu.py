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


class _FakeSolver:
    """Stand-in for SlabSolver: `plan` says which transports fail to initialise or to step."""
    plan = {}
    made = []

    def __init__(self, p, rank, world, transport):
        self.transport, self.steps, self.closed = transport, 0, False
        _FakeSolver.made.append(self)

    def Initialize(self):
        if _FakeSolver.plan.get(self.transport) == "init":
            raise RuntimeError("no such transport here")
        return self

    def step(self, n=1):
        if _FakeSolver.plan.get(self.transport) == "step":
            raise RuntimeError("the transfer did not finish")
        self.steps += n

    def sync(self):
        pass

    def close(self):
        self.closed = True


@pytest.mark.parametrize("plan,want,transport,warm_left,abandoned", [
    ({}, "rccl", "rccl", 2, 0),                       # the trial step is the first warm-up step
    ({"rccl": "init"}, "rccl", "host", 3, 0),         # RCCL cannot be set up: host-staged halos
    ({"rccl": "step"}, "rccl", "host", 3, 1),         # a transfer times out: abandoned, never closed
    ({}, "host", "host", 3, 0),
])
def test_bench_transport_selection(plan, want, transport, warm_left, abandoned, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "ABANDONED", [])
    _FakeSolver.plan, _FakeSolver.made = plan, []
    s, t, w = bench.open_slab_solver(_FakeSolver, None, 0, 2, want, 3, lambda v: v)
    assert (t, w, len(bench.ABANDONED)) == (transport, warm_left, abandoned)
    assert s.transport == transport and not s.closed
    for other in _FakeSolver.made:
        assert other is s or not other.closed      # failed to initialise, or abandoned: never closed
    # RCCL fine here but not on another rank: this rank's RCCL solver is closed, all use the host transport
    _FakeSolver.plan, _FakeSolver.made = {}, []
    s, t, w = bench.open_slab_solver(_FakeSolver, None, 0, 2, "rccl", 3, lambda v: 0)
    if want == "rccl":
        assert t == "host" and _FakeSolver.made[0].closed and s is _FakeSolver.made[1]
    # without warm-up steps there is no trial
    _FakeSolver.plan, _FakeSolver.made = {"rccl": "step"}, []
    s, t, w = bench.open_slab_solver(_FakeSolver, None, 0, 2, "rccl", 0, lambda v: v)
    assert (t, w) == ("rccl", 0)
