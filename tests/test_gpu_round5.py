"""Round 5 GPU tests (VERDICT r04 "next round" items)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------- item 1: bench.py starts its own ranks
@pytest.mark.parametrize("n", [2, 8])
def test_bench_launches_its_own_ranks(n):
    """exactly the driver's command form -- `python3 bench.py --gpus N ...`, no torchrun in front -- must
    run N ranks (reference: run_quant.sh:15 starts them, quant.py:149-155 joins them).  On a box with fewer than N GPUs the
    ranks share the GPU over gloo (asked for explicitly); with N GPUs the same command without --backend runs RCCL."""
    backend = [] if torch.cuda.device_count() >= n else ["--backend", "gloo"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *backend, "--steps", "1", "--warmup", "1",
           "--workload", "tinyllama-block-q4k"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2  # rank 0 only: the full dict, then the compact headline
    full, compact = lines
    for l in lines:
        assert l["n_gpus"] == n and l["ranks_seen"] == n and l["value"] > 0
    assert len(p.stdout.splitlines()[-1]) <= 2048 and p.stdout.splitlines()[-1].startswith("{")
    assert full["collectives_per_step"]["all_gather"] == 1 and full["allreduce_probe"]["ms"] > 0
    assert compact["roofline"]["frac"] > 0 and compact["cpu_baseline"]["value"] > 0
    assert full["config"]["calib_seqs_per_rank"] == 32 // n


def test_bench_refuses_more_ranks_than_gpus_without_gloo():
    """`--gpus N` on a box with fewer GPUs and no explicit gloo: refuse (non-zero exit, nothing measured) instead of
    printing an n_gpus = 1 line."""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "RCCL needs one GPU per rank" in p.stderr and "{" not in p.stdout


def test_bench_rank_mismatch_is_fatal():
    """A launcher that started ONE rank for `--gpus 2` must not produce a line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "tinyllama-block-q4k"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "rank(s) answered" in p.stderr and "{" not in p.stdout
