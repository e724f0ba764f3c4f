#!/usr/bin/env python
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc ... -- python bench.py --steps 1
--warmup 0` (tools/rocpd_stats.py output of each pass):

    python tools/make_traffic_json.py <fetch_stats.txt> <write_stats.txt> <rows> [tag]

HBM bytes per input row of k_part_scatter and k_part_aggregate = (FETCH_SIZE x 2 [gfx950: wide coalesced reads are
tallied at half, MI355X_MICROARCH.md] + WRITE_SIZE) KB x 1024 / rows, summed over the launches of one step; the
hash of heavydb_amd/csrc/kernels_part.hip is recorded so that bench.py only quotes the counters for the build they
were measured on."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sums(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\S+?)(?:<.*?>)?[\s(].*?%s\s+n=\s*(\d+)\s+sum=([0-9.e+]+)" % counter, line)
        if m:
            out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(3))
    return out


def main():
    fetch, write, rows = sys.argv[1], sys.argv[2], float(sys.argv[3])
    tag = sys.argv[4] if len(sys.argv) > 4 else "r03"
    f, w = sums(fetch, "FETCH_SIZE"), sums(write, "WRITE_SIZE")
    src = os.path.join(ROOT, "heavydb_amd", "csrc", "kernels_part.hip")
    out = {"_comment": "HBM bytes per input row of the dominant kernels of the default workload (cfg3f, 10 B rows), from separate "
                       "rocprofv3 --pmc passes of `python bench.py --steps 1 --warmup 0`: (FETCH_SIZE x 2 + WRITE_SIZE) KB x 1024 / rows",
           "kernel_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]}
    for k in ("k_part_scatter", "k_part_aggregate"):
        fk = sum(v for n, v in f.items() if n.startswith(k))
        wk = sum(v for n, v in w.items() if n.startswith(k))
        out[k] = {"hbm_bytes_per_row": round((2 * fk + wk) * 1024 / rows, 2),
                  "from": f"{tag}: (2 x {fk:.6g} [FETCH_SIZE KB] + {wk:.6g} [WRITE_SIZE KB]) x 1024 / {rows:.3g} rows"}
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
