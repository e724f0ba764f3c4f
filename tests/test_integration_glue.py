"""integration/: the HeavyDB-side binding (Mi355qExecutor.cpp — RelAlgExecutionUnit / FetchResult in, ResultSet
out, at the seam run_query_external uses) as a real translation unit, compiled against minimal mock HeavyDB
headers and driven by a native program that checks the ResultSetStorage-layout buffer against the oracle
(ExternalExecutor.cpp:420-530).  CPU: it builds and links; GPU: it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "integration", "glue_check")


def _build():
    r = subprocess.run(["bash", os.path.join(ROOT, "integration", "build.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(BIN)


def test_binding_translation_unit_compiles_without_hip():
    """the binding itself is plain C++17 over the C-ABI header: g++ -fsyntax-only, no HIP toolchain"""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DMI355Q_GLUE_MOCK_HEADERS",
                        "-I" + os.path.join(ROOT, "integration"), "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "integration", "Mi355qExecutor.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_glue_check_builds_and_links():
    _build()
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([BIN], capture_output=True, text=True)
        assert r.returncode == 77, r.stdout + r.stderr   # "no GPU": the program loaded both libraries and started


@pytest.mark.gpu
def test_glue_check_runs_on_the_device():
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all queries agree with the oracle" in r.stdout
