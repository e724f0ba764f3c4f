"""In-tree build of libmi355q.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m heavydb_amd._build          # incremental
    python -m heavydb_amd._build --force

The shared object lands in heavydb_amd/lib/ (git-ignored, but it travels to the GPU box with
the gpurun snapshot).  No torch extension machinery: the library is a plain C-ABI .so.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmi355q.so")

SOURCES = ["api.cpp", "api_projection.cpp", "api_result.cpp", "api_join.cpp", "boolfilter.cpp", "plan.cpp", "kernels_generic.hip", "kernels_fast.hip", "kernels_part.hip", "kernels_sort.hip", "kernels_lds.hip", "kernels_idx.hip", "kernels_proj.hip", "kernels_filter.hip"]  # missing files are skipped
HEADERS = ["dev_common.h", "rowfunc.h", "plan.h", "kernels.h", "fast_common.h", "expr.h", "lds_args.h", "api_internal.h", "boolfilter.h", "regprog.h", os.path.join("..", "..", "include", "mi355q.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for p in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.sep not in p or os.path.exists(p):
            return p
    return "hipcc"


def _newest_header_mtime() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _newest_header_mtime()
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
             "-fno-gpu-rdc"]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJDIR, src + ".o")
        objs.append(op)
        stale = force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_m)
        if stale:
            cmd = [hipcc] + flags + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for cmd, r in ex.map(run, jobs):
                if verbose or r.returncode != 0:
                    sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed for " + cmd[-3])
    need_link = force or bool(jobs) or not os.path.exists(LIB)
    if need_link:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
