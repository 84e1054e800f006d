"""bench.py's own launcher: `python bench.py --gpus N` with no launcher around it starts its N
rank processes itself (the driver invokes it exactly like that)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.parametrize("n", [2, 4])
def test_self_launch_rendezvous_on_cpu(n):
    """The launcher alone, on CPU: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    a gloo rendezvous on 127.0.0.1 among them, rank 0's line relayed on the parent's stdout,
    exit code 0.  (The GPU work of bench.py is replaced by the probe; the process tree, the
    environment and the relay are the real ones.)"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n)], capture_output=True, text=True, timeout=600,
                       env=_clean_env(SCHWZ_BENCH_LAUNCH_PROBE="1"))
    assert p.returncode == 0, p.stdout + p.stderr
    got = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert got["launch_probe"] and got["world"] == n and got["n_gpus"] == n
    assert got["ranks"] == [[r, r] for r in range(n)]


def test_self_launch_reports_a_failing_rank():
    """A rank that dies makes the launcher exit non-zero (here: every rank rejects a bad flag)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-such-flag"], capture_output=True, text=True,
                       timeout=300, env=_clean_env(SCHWZ_BENCH_LAUNCH_PROBE="1"))
    assert p.returncode != 0


def test_self_launch_ends_the_other_ranks_when_one_dies_at_start_up():
    """Rank 1 exits before the rendezvous: rank 0 would wait for it until the process-group timeout.  The
    launcher watches every child, ends the survivors and reports the failing rank's code within seconds."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=240,
                       env=_clean_env(SCHWZ_BENCH_LAUNCH_PROBE="1", SCHWZ_BENCH_PROBE_FAIL_RANK="1"))
    assert p.returncode == 3, p.stdout + p.stderr
    assert time.time() - t0 < 120
    assert "other ranks were stopped" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_itself_and_prints_one_line():
    """`python bench.py --gpus 2` exactly as the driver calls it.  On a box with fewer GPUs than
    ranks the two ranks share the GPU and stage halos through host over gloo (the line says so);
    with two GPUs the same command runs on RCCL.  Small slabs: this checks the launch path, the
    distributed branch and the shape of the line, not performance."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--slab", "48,40,16",
                        "--no-cpu-baseline", "--ttr-budget-s", "20", "--strong-grid", "48,40,32", "--strong-steps", "3"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and got["steps"] == 3 and got["scaling"] == "weak"
    assert got["unit"] == "subdomain-iter/s" and got["value"] > 0
    assert got["config"]["rows_per_gpu"] >= 48 * 40 * 16
    import torch
    backend = got["config"]["exchange_backend"]
    assert backend.startswith("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert 0.0 < got["roofline"]["frac"] <= 1.0
    # the strong-scaling leg of the same job: the fixed grid cut into N z-slabs
    assert got["strong"]["grid"] == "48x40x32" and got["strong"]["subdomains"] == 2 and got["strong"]["outer_iter_per_s"] > 0
    assert "mirror_bench_ras" not in got   # the C++ mirror's line belongs to the N = 1 run
    assert got["time_to_residual_converged"] and got["true_relative_residual"] < 1e-4
