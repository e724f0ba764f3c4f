#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database: per-kernel count / total / avg / min / max
duration (like --stats), optionally PMC counter sums per kernel.  usage: rocpd_stats.py DB"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def tab(prefix):
        return next(t for t in tabs if t.startswith(prefix))
    kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[1])
    q = f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) " \
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc"
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, n, t, a, mn, mx in rows:
        short = name.replace("void ", "").replace("mq::(anonymous namespace)::", "").replace("mq::fast::", "")
        short = re.sub(r"\(mq::.*|\(signed char.*|\(long\*.*|\(void\*.*", "", short)[:70]
        print(f"{short:70s} {n:6d} {t/1e6:10.3f} {a/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f} {100*t/tot:6.1f}")
    try:
        pe, pi = tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
        q = f"select s.{namecol}, i.name, count(*), sum(p.value) from {pe} p join {pi} i on p.pmc_id = i.id " \
            f"join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by 1, 2 order by 1, 2"
        prow = list(cur.execute(q))
        if prow:
            print("\nPMC (sum over dispatches, and per dispatch):")
            for name, cname, n, v in prow:
                short = name.replace("void ", "").replace("mq::(anonymous namespace)::", "").replace("mq::fast::", "")
                short = re.sub(r"\(mq::.*|\(signed char.*|\(long\*.*|\(void\*.*", "", short)[:60]
                print(f"{short:60s} {cname:24s} n={n:5d} sum={v:.6g} per_dispatch={v/max(n,1):.6g}")
    except Exception as e:  # no counters collected
        pass


if __name__ == "__main__":
    main(sys.argv[1])
