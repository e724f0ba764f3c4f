"""The reference's synthetic benchmark queries (tools/refbench.py: Benchmarks/synthetic_benchmark/queries/*/*.sql on
create_table.py's schema) through the HIP library against the oracle: at the CPU tests' small size with the
plan-time kernel choice and with the large-input family members forced (kernel_variant 2), and at 16 M rows — large
enough for the planner to pick the partitioned / packed / projected routes itself — for one query per family."""
import os
import sys

import numpy as np
import pytest

from heavydb_amd import capi
from tests.helpers import compare_buffers, compare_rows

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refbench  # noqa: E402

QUERIES = refbench.queries()


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


def _run(torch, oracle, name, n_rows, card_cap, variant, flags=0, report=None):
    from heavydb_amd.executor import Executor, FetchResult
    names, descs, gens = refbench.schema(card_cap)
    cols = [oracle.generate_column(n_rows, g[0], g[1], g[2], g[3], g[4], g[5]) for g in gens]
    cut = (n_rows // 3) // 4 * 4
    frags = [[c[:cut] for c in cols], [c[cut:] for c in cols]]
    ra, _ = refbench.build_unit(QUERIES[name], names, descs, n_rows)
    q, want, code = oracle.execute(ra.to_plan(), frags, n_threads=min(os.cpu_count() or 1, 16))
    assert code == 0
    dev = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]
    es = [t.element_size() for t in dev]
    fr = FetchResult([[int(t.data_ptr()) for t in dev], [int(t.data_ptr()) + cut * e for t, e in zip(dev, es)]],
                     [cut, n_rows - cut], keepalive=dev)
    rs = Executor(0).executeWorkUnit(ra, fr, kernel_variant=variant, flags=flags)
    compare_buffers(q, want, rs.getStorage(), 1e-9)
    compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), 1e-9)
    if report is not None:
        report["variant"] = int(rs.report.variant)
    return rs.report.kernel_name.decode()


@pytest.mark.parametrize("variant", [0, 2], ids=["planned", "large_input_members"])
@pytest.mark.parametrize("name", list(QUERIES), ids=list(QUERIES))
def test_refbench_queries_small(torch_cuda, oracle, name, variant):
    _run(torch_cuda, oracle, name, 30_000, 3_000, variant)


@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("name", ["PHS004", "PHM003", "BH004", "BH007", "MSPHS002", "MSPHM002", "MSBS002"])
def test_refbench_windowed_lds_groupby(torch_cuda, oracle, name, member):
    """the 10 K-group shapes at their real cardinality: tables that do not fit one LDS run as 2 - 8 windows of
    k_groupby_lds (perfect hash: ranges of the entry index; baseline, FLOAT / DOUBLE / BIGINT key: classes of a key hash),
    through the typed member (roles compiled in, variant 5) and the run-time-role member (variant 4)"""
    rep = {}
    kernel = _run(torch_cuda, oracle, name, 400_000, 10_000, 0,
                  flags=capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0, report=rep)
    assert kernel == "k_groupby_lds", kernel
    assert rep["variant"] == (4 if member == "generic" else 5), rep


@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("name", ["PHS001", "PHS002", "PHS003", "PHM001", "PHM002", "BH001", "BH002", "BH003", "MSBS001",
                                  "MSPHS001", "MSPHM001"])
def test_refbench_small_lds_groupby_both_members(torch_cuda, oracle, name, member):
    """the few-groups shapes (table replicated K times per workgroup) at 3 M rows (odd count: quad remainder and tail rows),
    every member of the typed family the benchmark reaches against the oracle"""
    rep = {}
    kernel = _run(torch_cuda, oracle, name, 3_000_003, 0, 0,
                  flags=capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0, report=rep)
    if kernel == "k_groupby_lds":
        assert rep["variant"] == (4 if member == "generic" else 5), rep


@pytest.mark.parametrize("name", ["NGA02", "NGA05", "PHS003", "PHS004", "PHS006", "PHM004", "BH002", "BH004", "BH006",
                                  "BH008", "BH010", "MSBS001", "MSBS003", "MSPHS002", "MSPHM006", "S002"])
def test_refbench_queries_16m_rows(torch_cuda, oracle, name):
    kernel = _run(torch_cuda, oracle, name, 16_000_000, 0, 0)
    print(name, kernel)


@pytest.mark.parametrize("name", ["PHS005", "PHS006", "PHM004", "PHM005", "MSPHS003", "MSPHS006", "MSPHS007", "MSPHS008",
                                  "MSPHS010", "MSPHM003", "MSPHM006"])
def test_refbench_idx_partitioned_family(torch_cuda, oracle, name):
    """perfect-hash tables too large for LDS (100 K+ entries, 1 - 3 INT keys, 1 - 3 INT value columns): k_idx_scatter +
    k_idx_aggregate (kernels_idx.hip; 8-byte records for one value column, 16-byte records for two or three, one exchange)
    against the oracle; kernel_variant 2 takes the family on a 2 M-row input (odd count: quad remainders and tails)"""
    rep = {}
    kernel = _run(torch_cuda, oracle, name, 2_000_003, 150_000, 2, report=rep)
    # (round 6: the benchmark's value columns carry their ranges — the packed 2- / 4-byte word, variants 8 / 7)
    assert kernel == "k_idx_scatter" and rep["variant"] in (7, 8), (kernel, rep)
    kernel = _run(torch_cuda, oracle, name, 2_000_003, 150_000, 2, flags=capi.OPT_NO_IDX_PACK, report=rep)
    assert kernel == "k_idx_scatter" and rep["variant"] == 6, (kernel, rep)


@pytest.mark.parametrize("name", ["S001", "S003", "PHS007", "MSPHS005", "MSPHS012", "PHM006", "MSPHM005"])
def test_refbench_idx_partitioned_family_packed_at_benchmark_cardinalities(torch_cuda, oracle, name):
    """the packed word at the benchmark's own cardinalities (10 M entries: 14 index bits + the value codes in 4 bytes; the
    Sort shapes: the index alone in 2 bytes), 16 M rows against the oracle"""
    rep = {}
    kernel = _run(torch_cuda, oracle, name, 16_000_003, 0, 0, report=rep)
    assert kernel == "k_idx_scatter" and rep["variant"] in (7, 8), (name, kernel, rep)


def test_refbench_idx_partitioned_family_planned_at_16m_rows(torch_cuda, oracle):
    """at the benchmark's own cardinalities the planner picks the family itself from 8 M rows on"""
    for name in ["PHS007", "MSPHS005", "MSPHS012"]:
        rep = {}
        kernel = _run(torch_cuda, oracle, name, 16_000_000, 0, 0, report=rep)
        assert kernel == "k_idx_scatter", (name, kernel)
