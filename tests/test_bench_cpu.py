"""bench.py host logic that needs no GPU: `python bench.py --gpus N` started without a launcher
becomes the launcher of N fresh ranks (before anything touches torch or the GPU), and the counter
summary is only reported for the kernel sources it was measured on."""
import json
import os
import subprocess
import sys

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
