"""bench.py --gpus N from a bare shell: the self-launcher starts one process per GPU under torch.distributed.run with the
driver's contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1). Runs on CPU: the ranks only report what
they were started with (TBRM_BENCH_LAUNCH_CHECK=1) and never touch a device."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TBRM_BENCH_LAUNCH_CHECK"] = "1"
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return [json.loads(obj) for obj in re.findall(r"\{[^{}]*\}", out.stdout)]  # (two ranks may share a line)


def test_one_gpu_needs_no_launcher():
    (r,) = run_bench([])
    assert r["world_size"] == 1 and r["rank"] == 0 and r["config"] == 3  # the metric's config at N = 1


def test_gpus_2_relaunches_itself_under_torchrun():
    ranks = run_bench(["--gpus", "2", "--steps", "3"])
    assert sorted(r["rank"] for r in ranks) == [0, 1]
    assert all(r["world_size"] == 2 and r["gpus"] == 2 and r["master_addr"] == "127.0.0.1" for r in ranks)
    assert all(r["config"] == 3 for r in ranks)  # the SAME workload as N = 1: value(N) / value(1) is a speed-up


def test_config_5_is_opt_in_at_n_gpus():
    ranks = run_bench(["--gpus", "2", "--config", "5"])
    assert all(r["config"] == 5 for r in ranks)  # north_star's tile-scaling workload (one 2048^2 frame), on request


def test_already_launched_ranks_do_not_relaunch():
    (r,) = run_bench(["--gpus", "2"], {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r["rank"] == 1 and r["world_size"] == 2
