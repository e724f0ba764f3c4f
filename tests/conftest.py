import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")) as f:
        return json.load(f)


# ---- MI355Q_HOSTSIM=1: run the `-m gpu` tests on a machine WITHOUT a GPU against the host simulation of the library
# (tests/hostsim: api.cpp / plan.cpp / kernels_generic.hip on a stand-in HIP runtime, row-function stand-ins for the
# fast kernel families).  A pre-flight for the host code paths the gpu tests drive — it says nothing about the fast
# kernels themselves, and tests that need device-only members (ORDER BY, payload probes, slice merges) or sizes a CPU
# cannot visit fail or time out here by design.  Usage:
#   MI355Q_HOSTSIM=1 python -m pytest tests -m gpu -p no:cacheprovider --timeout 120 -q
# MI355Q_HOSTSIM=real: the same with the REAL kernels_fast.hip / kernels_lds.hip compiled for the host (hostsim_lib(real_fast=True)).
# Never set on a GPU box: the driver's gpu run loads the real libmi355q.so.
if os.environ.get("MI355Q_HOSTSIM") in ("1", "real"):
    @pytest.fixture(scope="session", autouse=True)
    def _hostsim_session():
        import torch
        from heavydb_amd import capi
        from tests.helpers import hostsim_lib
        assert not torch.cuda.is_available(), "MI355Q_HOSTSIM is for machines without a GPU"
        lib = capi.load_library(hostsim_lib(real_fast=os.environ.get("MI355Q_HOSTSIM") == "real"))
        capi._lib = lib
        real_zeros, real_empty, real_full, real_arange = torch.zeros, torch.empty, torch.full, torch.arange
        strip = lambda f: (lambda *a, **k: f(*a, **{x: y for x, y in k.items() if x != "device"}))  # noqa: E731
        torch.zeros, torch.empty, torch.full, torch.arange = strip(real_zeros), strip(real_empty), strip(real_full), strip(real_arange)
        torch.Tensor.cuda = lambda self, *a, **k: self.clone()   # a "device" copy: it must not alias the host array it came from
        torch.cuda.is_available = lambda: True
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.device_count = lambda: 1
        yield
