#!/usr/bin/env python
"""Replays tests/test_gpu_parity.py::test_hip_matches_oracle_on_fuzzed_row_plans for given seeds on the GPU and
prints the plan of every iteration that fails (instead of stopping at the first): gpu_fuzz_replay.py <iters> <seed>..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from heavydb_amd import capi
    from heavydb_amd.executor import Executor, FetchResult
    from oracle import oracle
    from tests.helpers import compare_buffers, compare_rows
    from tests.test_plan_fuzz import _fuzz_row_plan, _fuzz_table
    iters = int(sys.argv[1])
    for seed in [int(a) for a in sys.argv[2:]]:
        rng = np.random.default_rng(seed)
        ex = Executor(0)
        for i in range(iters):
            n_rows = int(rng.integers(8, 3000))
            descs, cols = _fuzz_table(rng, n_rows)
            ra = _fuzz_row_plan(rng, descs)
            cut = (n_rows // 2) & ~3
            frags = [[c[:cut] for c in cols], [c[cut:] for c in cols]]
            try:
                q, want, code = oracle.execute(ra.to_plan(), frags, n_threads=2)
            except capi.Mi355qError:
                continue
            if code != 0:
                continue
            dev = [[torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in f] for f in frags]
            fr = FetchResult([[int(t.data_ptr()) for t in f] for f in dev], [len(f[0]) for f in frags], keepalive=dev)
            variant, fg = int(rng.integers(0, 3)), bool(rng.integers(0, 5) == 0)

            def describe(what):
                print(f"seed {seed} iteration {i}: {what}; variant {variant} force_generic {fg} rows {n_rows}")
                print("  quals", [(x.col, x.op, x.literal) for x in ra.simple_quals], "group", ra.groupby_exprs,
                      "targets", [(t.agg, t.col, (t.cond.col, t.cond.op, t.cond.literal) if getattr(t, "cond", None) else None) for t in ra.target_exprs],
                      "bigint_count", ra.bigint_count)
                for j, d in enumerate(descs):
                    print("  col", j, "type", d.type, "nullable", d.nullable, "enc", d.encoding, "logical", d.logical_type,
                          "range", d.range.valid, d.range.min, d.range.max, d.range.has_nulls, d.range.bucket)
                print("  desc", q.desc_type, "keyless", q.keyless, "idx_key", q.idx_target_as_key, "entries", q.entry_count,
                      "slot_width", q.slot_width, "init", [q.init_vals[k] for k in range(q.slot_count)],
                      "skip", [q.target_skip_null[t] for t in range(q.n_targets)], "slots", [q.target_slot[t] for t in range(q.n_targets)])
            try:
                rs = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=variant, force_generic=fg)
            except capi.Mi355qError as e:
                describe(f"execute raised {e}")
                continue
            key_t = [t for t in range(q.n_targets) if q.keyless and
                     (q.target_slot[t] == q.idx_target_as_key or
                      (q.target_agg[t] == capi.AVG and q.target_slot[t] == q.idx_target_as_key - 1))]
            if key_t and q.target_skip_null[key_t[0]]:
                continue
            try:
                compare_buffers(q, want, rs.getStorage(), 1e-9)
                compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), 1e-9)
            except AssertionError as e:
                describe(f"mismatch in {rs.report.kernel_name.decode()}: {str(e)[:400]}")
                wi, wd, wn = oracle.fetch_rows(q, want)
                gi, gd, gn = rs.fetch()
                print("  want rows", wi.shape, "got rows", gi.shape)
                if wi.shape == gi.shape:
                    bad = np.argwhere(wn != gn)[:4]
                    print("  null flag differences at", bad.tolist(), "want", [(wi[r].tolist(), wn[r].tolist()) for r, _ in bad[:2]],
                          "got", [(gi[r].tolist(), gn[r].tolist()) for r, _ in bad[:2]])


if __name__ == "__main__":
    main()
