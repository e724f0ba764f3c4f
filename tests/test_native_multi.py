"""tools/native/multi_check: INTEGRATION.md section 5 as a native program — per-device C-ABI calls
(mi355q_execute_async, mi355q_shard_pads, mi355q_shard_merge_slices) with RCCL called directly (grouped
ncclSend / ncclRecv, ncclAllReduce), no Python, no torch.  CPU: it compiles and links against librccl and the
library; GPU: it runs on as many devices as the box has (1 here: the collectives degenerate, every C-ABI call of
the sequence still executes on the device)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "native", "multi_check")


def _build():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "native", "build_multi_check.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(BIN)


def test_multi_check_builds_and_links_against_rccl():
    _build()
    ldd = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "libmi355q" in ldd
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([BIN], capture_output=True, text=True)
        assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
def test_multi_check_runs():
    if not os.path.exists(BIN):
        _build()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([BIN, "8", "128e6", "1e6"], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "multi_check: ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [3, 8])
def test_multi_check_runs_with_simulated_ranks(ranks):
    """`multi_check --sim N`: N ranks on ONE device (own threads, streams, fragments f % N, tables), the grouped send / recv of
    the slice and pad exchanges as device-to-device copies behind stream events: the N x N slices really move and every rank
    folds N sources with mi355q_shard_merge_slices — what RCCL on a 1-GPU box degenerates to nothing."""
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN, "--sim", str(ranks), "192e6", "1e6"], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "multi_check: ok" in r.stdout and f"{ranks} rank(s) simulated" in r.stdout
