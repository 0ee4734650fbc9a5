"""bench.py on a GPU: the one JSON line of a single-GPU run and of a self-launched 2-rank run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def run_bench(args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout          # ONE line on stdout, everything else goes to stderr
    return json.loads(lines[0]), r.stderr


def test_single_gpu_line_has_the_contract_fields():
    out, _ = run_bench(["--size", "128", "--iters", "16", "--steps", "3", "--warmup", "1", "--cpu-sample", "32"])
    assert out["n_gpus"] == 1 and out["scaling"] is None and out["unit"] == "cell-updates/s"
    assert out["value"] == pytest.approx(128 ** 3 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0


def test_two_ranks_launched_by_bench_itself():
    """`python bench.py --gpus 2` without a launcher: bench.py starts its two ranks.  On a box with
    one GPU both share it, RCCL refuses that and every rank falls back to host-staged halos."""
    out, err = run_bench(["--gpus", "2", "--size", "64", "--iters", "8", "--steps", "2", "--warmup", "2"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["nz"] == 128
    assert out["value"] == pytest.approx(64 * 64 * 128 * 2 / (out["ms_per_step"] * 2e-3), rel=1e-6)
    halos = out["config"]["workload"]
    assert ("RCCL" in halos and out["rccl_ranks"] == 2) or ("host buffers" in halos and out["rccl_ranks"] == 0), halos


def test_two_ranks_strong_scaling_and_convergence_mode():
    """The other multi-rank shapes of the benchmark: one grid cut into slabs (--scaling strong), and
    the weak-scaling column swept until the all-reduced residual norm meets a tolerance."""
    out, _ = run_bench(["--gpus", "2", "--size", "64", "--iters", "8", "--steps", "2", "--warmup", "1",
                        "--scaling", "strong"])
    assert out["scaling"] == "strong" and out["config"]["nz"] == 64 and out["n_gpus"] == 2
    assert out["value"] == pytest.approx(64 ** 3 * 2 / (out["ms_per_step"] * 2e-3), rel=1e-6)
    out, _ = run_bench(["--gpus", "2", "--size", "32", "--iters", "64", "--steps", "2", "--warmup", "1",
                        "--tolerance", "1e-2", "--check-every", "8"])
    assert out["scaling"] == "weak" and out["config"]["nz"] == 64
    assert out["config"]["tolerance"] == 1e-2 and 8 <= out["config"]["sweeps_per_step_run"] <= 64
