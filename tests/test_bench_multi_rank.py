"""`python bench.py --gpus 2` without a launcher starts its two ranks itself (python -m torch.distributed.run, 127.0.0.1)
and prints one JSON line from rank 0.  Run here with --hostsim: every rank drives the host simulation of the library
(the real device sources compiled for the CPU) on a tiny table and gloo carries the final merge — the N-rank control flow
of the script (rank environment, fragment dealing, barrier + max-over-ranks timing, merge, per-rank breakdown) end to end
on a machine without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,rows", [("cfg3f", 2_000_000), ("cfg2", 1_000_000)])
def test_bench_self_launches_two_ranks(config, rows):
    from tests.helpers import hostsim_lib
    hostsim_lib(real_fast=True)  # built once, before the ranks race for it
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", config,
                        "--rows", str(rows), "--hostsim", "--no-cpu-baseline", "--verify"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_reported_by_rccl"] == 2 and d["steps"] == 1
    assert len(d["per_rank"]) == 2 and sorted(p["rank"] for p in d["per_rank"]) == [0, 1]
    assert sum(p["rows"] for p in d["per_rank"]) == rows
    assert d["verify"]["groups"] >= 1
    assert "HOST SIMULATION" in d["data"]


def test_bench_refuses_a_launcher_of_the_wrong_size():
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--hostsim"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "launcher started 1 rank" in r.stderr
