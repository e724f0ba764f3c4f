#!/usr/bin/env python
"""Markdown tables for profiles/README.md from the round's committed lines:

    python tools/make_profiles_tables.py r04 profiles/r03_refbench_1b_call6.jsonl

prints (1) one row per profiles/<round>_bench_*.json — config, ms per step, dominant kernel's roofline fraction, whole-step fraction, CPU baseline,
GPU / CPU — and (2) the reference's synthetic benchmark at 1 B rows of this round next to the previous round's table."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_lines(path):
    out = {}
    for l in open(path):
        d = json.loads(l)
        if "query" in d:
            out[d["query"]] = d
    return out


def main():
    rnd = sys.argv[1]
    prev = load_lines(sys.argv[2]) if len(sys.argv) > 2 else {}
    print("| file | workload | ms / step | value | dominant kernel: achieved / frac | whole step | traffic (PMC) | cpu_baseline | verify |")
    print("|---|---|---|---|---|---|---|---|---|")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_bench_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        r, c = d.get("roofline", {}), d.get("cpu_baseline") or {}
        print("| `%s` | %s | %.3f | %.3g %s | `%s` %.0f GB/s / %.3f | %s | %s | %s | %s |" % (
            os.path.basename(f), d["config"]["workload"], d["ms_per_step"], d["value"], d["unit"], r.get("kernel"), r.get("achieved", 0),
            r.get("frac", 0), "%.3f" % r["whole_step_frac"] if r.get("whole_step_frac") else "—",
            "%.3g B/launch" % r["traffic"] if r.get("traffic") else "—",
            "%.3g %s, %s cores (%s)" % (c["value"], c["unit"], c["cores"], c["kind"]) if c else "—",
            "ok" if d.get("verify") else "—"))
    cur_path = os.path.join(ROOT, "profiles", f"{rnd}_refbench_1b_final.jsonl")
    if os.path.exists(cur_path):
        cur = load_lines(cur_path)
        print()
        print("| query | route (`mi355q_explain`) | whole step | of 8 TB/s | previous round |")
        print("|---|---|---|---|---|")
        for q, d in cur.items():
            p = prev.get(q, {})
            print("| %s | %s | %.2f ms | %.3f | %s |" % (q, (d.get("route") or d.get("kernel") or "").replace("|", "\\|"), d.get("ms", 0),
                                                      d.get("whole_step_frac", 0), "%.1f ms" % p["ms"] if p.get("ms") else "—"))


if __name__ == "__main__":
    main()
