"""Comparison helpers for ResultSetStorage buffers (oracle vs product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from heavydb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMPTY64 = 2**63 - 1
EMPTY32 = 2**31 - 1


def qmd_equal(a: capi.QMD, b: capi.QMD):
    da, db = a.as_dict(), b.as_dict()
    assert da == db, {k: (da[k], db[k]) for k in da if da[k] != db[k]}


def fp_slots(q: capi.QMD):
    """Slot indices holding double bits."""
    s = set()
    for t in range(q.n_targets):
        if q.target_arg_is_fp[t] and q.target_slot[t] >= 0:
            s.add(q.target_slot[t])
    return s


def f32_slots(q: capi.QMD):
    """Slot indices holding FLOAT bits in their low 4 bytes (aggregates of a float column)."""
    s = set()
    for t in range(q.n_targets):
        if q.target_arg_is_f32[t] and q.target_slot[t] >= 0:
            s.add(q.target_slot[t])
    return s


F32_RTOL = 2e-4  # single-precision accumulation: the order of the additions differs between runs
F32_ATOL = 0.05  # ... and sums that cancel (values of either sign, |v| <= 100, a few thousand rows)
                 # keep an absolute error of about n * max|v| * 2^-24 whatever their own size


def _f32_equalise(q: capi.QMD, want: np.ndarray, got: np.ndarray, kq: int):
    """Float slots are compared as floats (upper 4 bytes exactly, low 4 bytes within F32_RTOL);
    where they agree the `got` quad is overwritten with the `want` quad so the exact integer
    comparison that follows passes."""
    for s in f32_slots(q):
        w = np.ascontiguousarray(want[:, kq + s]).view(np.int32).reshape(-1, 2)
        g = np.ascontiguousarray(got[:, kq + s]).view(np.int32).reshape(-1, 2)
        assert (w[:, 1] == g[:, 1]).all(), "upper half of a float slot changed"
        wf, gf = w[:, 0].copy().view(np.float32), g[:, 0].copy().view(np.float32)
        ok = (w[:, 0] == g[:, 0]) | (np.isfinite(wf) & np.isfinite(gf) &
                                     (np.abs(wf - gf) <= F32_RTOL * np.maximum(np.abs(wf), np.abs(gf)) + F32_ATOL))
        assert ok.all(), (s, wf[~ok][:5], gf[~ok][:5])
        got[:, kq + s] = want[:, kq + s]


def _close(a: int, b: int, rtol: float) -> bool:
    if a == b:
        return True
    fa = np.array([a], dtype=np.int64).view(np.float64)[0]
    fb = np.array([b], dtype=np.int64).view(np.float64)[0]
    return bool(np.isfinite(fa) and np.isfinite(fb) and abs(fa - fb) <= rtol * max(abs(fa), abs(fb), 1e-300))


def key_matrix(q: capi.QMD, buf: np.ndarray) -> np.ndarray:
    """[entries, group_col_count] int64 matrix of the key components stored at the row starts."""
    kq = q.key_bytes // 8
    ng = max(q.group_col_count, 1)
    if q.key_width == 4:
        k32 = np.ascontiguousarray(buf[:, :kq]).view(np.int32).reshape(buf.shape[0], 2 * kq)
        return k32[:, :ng].astype(np.int64)
    return np.ascontiguousarray(buf[:, :ng])


def compare_buffers(q: capi.QMD, want: np.ndarray, got: np.ndarray, rtol: float = 1e-9):
    """Perfect-hash / non-grouped: index-aligned, bit-exact for integer quads, rtol on fp64
    slots.  Baseline: compared as key -> slots maps (slot positions are insertion-order
    dependent even in the reference; docs hash_joins.rst)."""
    rq, kq = q.row_size // 8, q.key_bytes // 8
    want = want.reshape(-1, rq)
    got = got.reshape(-1, rq)
    assert want.shape == got.shape
    fps = fp_slots(q)
    if q.desc_type == capi.GROUP_BY_BASELINE_HASH:
        def live_sorted(buf):
            keys = key_matrix(q, buf)
            live = keys[:, 0] != (EMPTY32 if q.key_width == 4 else EMPTY64)
            idx = np.nonzero(live)[0]
            k = keys[idx]
            order = np.lexsort(tuple(k[:, c] for c in range(k.shape[1])[::-1]))
            k = k[order]
            dup = np.nonzero((k[1:] == k[:-1]).all(axis=1))[0]
            assert dup.size == 0, f"duplicate key {k[dup[0]].tolist()} in table"
            return k, buf[idx][order][:, kq:]
        kw, sw = live_sorted(want)
        kg, sg = live_sorted(got)
        assert kw.shape == kg.shape and (kw == kg).all(), (kw.shape, kg.shape)
        sg = sg.copy()
        _f32_equalise(q, sw, sg, 0)
        kw = kw[:, 0] if kw.shape[1] == 1 else kw
        # 4-byte slots (COUNT(*) / key projections only) share quads two by two: compared quad by quad, exactly
        for s in range(sw.shape[1] if q.slot_width == 4 else q.slot_count):
            w, g = sw[:, s], sg[:, s]
            diff = np.nonzero(w != g)[0]
            if s in fps and q.slot_width == 8:
                fw, fg = w[diff].view(np.float64), g[diff].view(np.float64)
                ok = np.isfinite(fw) & np.isfinite(fg) & \
                    (np.abs(fw - fg) <= rtol * np.maximum(np.maximum(np.abs(fw), np.abs(fg)), 1e-300))
                assert ok.all(), (s, kw[diff[~ok]][:5], fw[~ok][:5], fg[~ok][:5])
            else:
                assert diff.size == 0, (s, kw[diff[:5]], w[diff[:5]], g[diff[:5]])
        return
    got = got.copy()
    _f32_equalise(q, want, got, kq)
    int_cols = [c for c in range(rq) if not (c >= kq and (c - kq) in fps)]
    bad = np.nonzero((want[:, int_cols] != got[:, int_cols]).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], want[bad[:5]], got[bad[:5]])
    for s in fps:
        w = want[:, kq + s]
        g = got[:, kq + s]
        diff = np.nonzero(w != g)[0]
        for i in diff:
            assert _close(int(w[i]), int(g[i]), rtol), (i, s, w[i], g[i])


def compare_rows(q: capi.QMD, want, got, rtol: float = 1e-9):
    """fetch_rows outputs (ival, dval, is_null): exact for ints/nulls, rtol for doubles;
    baseline rows are matched as multisets via sorting on the integer columns."""
    wi, wd, wn = want
    gi, gd, gn = got
    assert wi.shape == gi.shape, (wi.shape, gi.shape)
    if q.desc_type == capi.GROUP_BY_BASELINE_HASH and wi.shape[0] > 1:
        def order(i, d, nl):
            # integer columns first, then the NULL flags (a NULL reads as 0 in the value arrays: rows that tie
            # on every value may still differ in which of them is NULL); rows that tie on all of those (no
            # unique key projected) by the doubles rounded to single precision, i.e. well above the fp64 tolerance
            with np.errstate(over="ignore"):
                dr = np.where(np.isfinite(d), d, 0.0).astype(np.float32)
            # (last resort: the exact doubles — two rows closer than single precision but further apart than
            # the tolerance must not swap places between the two sides)
            dx = np.where(np.isfinite(d), d, 0.0)
            keys = ([dx[:, c] for c in range(d.shape[1])[::-1]] + [dr[:, c] for c in range(d.shape[1])[::-1]] +
                    [nl[:, c] for c in range(nl.shape[1])[::-1]] + [i[:, c] for c in range(i.shape[1])[::-1]])
            return np.lexsort(tuple(keys))
        ow, og = order(wi, wd, wn), order(gi, gd, gn)
        wi, wd, wn = wi[ow], wd[ow], wn[ow]
        gi, gd, gn = gi[og], gd[og], gn[og]
    assert (wi == gi).all()
    assert (wn == gn).all()
    frt = np.full(wd.shape[1], rtol)
    fat = np.zeros(wd.shape[1])
    for t in range(q.n_targets):
        if q.target_arg_is_f32[t]:
            frt[t] = max(rtol, F32_RTOL)
            fat[t] = F32_ATOL
    ok = (np.abs(wd - gd) <= frt[None, :] * np.maximum(np.abs(wd), np.abs(gd)) + fat[None, :]) | (wd == gd)
    assert ok.all(), (wd[~ok][:5], gd[~ok][:5])


def rowwise_qmd(q: capi.QMD) -> capi.QMD:
    r = capi.QMD.from_buffer_copy(q)
    r.output_columnar = 0
    return r


def columnar_to_rows(q: capi.QMD, flat: np.ndarray) -> np.ndarray:
    """The row images [entry_count, row_size / 8] of a columnar buffer — the layout written out
    from the reference's formulas a third time (numpy): group column g at g * 8 * E
    (getPrependedGroupColOffInBytes, none when keyless), then slot column s at
    s * align8(slot_width * E) (getColOffInBytes); total = getBufferSizeBytes."""
    assert q.output_columnar
    E = q.entry_count
    raw = np.ascontiguousarray(flat).view(np.int8).reshape(-1)
    kq = 0 if q.keyless else q.group_col_count
    assert kq == q.key_bytes // 8
    rows = np.zeros((E, q.row_size // 8), dtype=np.int64)
    off = 0
    for k in range(kq):
        rows[:, k] = raw[off:off + 8 * E].view(np.int64)
        off += 8 * E
    col_bytes = (q.slot_width * E + 7) // 8 * 8
    s32 = rows[:, kq:].view(np.int32) if q.slot_width == 4 else None
    for s in range(q.slot_count):
        if q.slot_width == 8:
            rows[:, kq + s] = raw[off:off + 8 * E].view(np.int64)
        else:
            s32[:, s] = raw[off:off + 4 * E].view(np.int32)
        off += col_bytes
    assert off == raw.nbytes, (off, raw.nbytes)
    return rows


def murmur3_u64(keys: np.ndarray) -> np.ndarray:
    """MurmurHash3_x86_32 of little-endian int64 keys, seed 0 (key_hash of one 8-byte key,
    GroupByRuntime.cpp:20-23 + MurmurHash3Inl.h) — numpy restatement, pinned against the
    reference-generated vectors in tests/golden (test_oracle_golden.py)."""
    k = keys.astype(np.int64).view(np.uint64)
    M = np.uint64(0xFFFFFFFF)

    def mul(a, b):
        return (a * np.uint64(b)) & M

    def rotl(x, r):
        return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M

    h = np.zeros(k.shape, dtype=np.uint64)
    for blk in (k & M, k >> np.uint64(32)):
        k1 = mul(blk, 0xcc9e2d51)
        k1 = mul(rotl(k1, 15), 0x1b873593)
        h ^= k1
        h = (mul(rotl(h, 13), 5) + np.uint64(0xe6546b64)) & M
    h ^= np.uint64(8)
    h ^= h >> np.uint64(16)
    h = mul(h, 0x85ebca6b)
    h ^= h >> np.uint64(13)
    h = mul(h, 0xc2b2ae35)
    h ^= h >> np.uint64(16)
    return h


def murmur3_words(words: np.ndarray) -> np.ndarray:
    """MurmurHash3_x86_32, seed 0, of rows of little-endian 4-byte blocks ([n, n_words] uint32):
    key_hash(key, key_count, key_width) for any key shape (GroupByRuntime.cpp:20-23)."""
    M = np.uint64(0xFFFFFFFF)
    w = words.astype(np.uint64)

    def mul(a, b):
        return (a * np.uint64(b)) & M

    def rotl(x, r):
        return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M

    h = np.zeros(w.shape[0], dtype=np.uint64)
    for i in range(w.shape[1]):
        k1 = mul(w[:, i], 0xcc9e2d51)
        k1 = mul(rotl(k1, 15), 0x1b873593)
        h ^= k1
        h = (mul(rotl(h, 13), 5) + np.uint64(0xe6546b64)) & M
    h ^= np.uint64(4 * w.shape[1])
    h ^= h >> np.uint64(16)
    h = mul(h, 0x85ebca6b)
    h ^= h >> np.uint64(13)
    h = mul(h, 0xc2b2ae35)
    h ^= h >> np.uint64(16)
    return h


def check_probe_invariant(q: capi.QMD, buf: np.ndarray):
    """A baseline buffer must be a valid image of get_group_value's linear probing
    (GroupByRuntime.cpp:25-48): every key sits at or after its home slot
    MurmurHash3(key bytes) % entry_count (cyclically) with no empty slot in between — i.e. the
    reference's own probe sequence finds it.  Any key shape (1..4 components, 4 or 8 bytes)."""
    if q.desc_type != capi.GROUP_BY_BASELINE_HASH:
        return
    rq, kq = q.row_size // 8, q.key_bytes // 8
    rows = buf.reshape(-1, rq)
    n = rows.shape[0]
    n_words = max(q.group_col_count, 1) * (q.key_width // 4)
    words = np.ascontiguousarray(rows[:, :kq]).view(np.uint32).reshape(n, 2 * kq)[:, :n_words]
    live = key_matrix(q, rows)[:, 0] != (EMPTY32 if q.key_width == 4 else EMPTY64)
    pos = np.nonzero(live)[0]
    if pos.size == 0:
        return
    home = (murmur3_words(words[pos]) % np.uint64(n)).astype(np.int64)
    # number of empty slots in the cyclic interval [home, pos) must be zero
    empties = np.concatenate([[0], np.cumsum(~live)]).astype(np.int64)  # empties[i] = # empty in [0, i)
    total_empty = int(empties[n])
    fwd = pos >= home
    gap = np.where(fwd, empties[pos] - empties[home], total_empty - empties[home] + empties[pos])
    bad = np.nonzero(gap != 0)[0]
    assert bad.size == 0, (bad.size, pos[bad[:5]], home[bad[:5]], rows[pos[bad[:5]], :kq])


_emu = None


def emu_lib() -> C.CDLL:
    """Host emulation of the product's row logic (tests/emu/emu.cpp, -DMQ_EMU)."""
    global _emu
    if _emu is None:
        out = os.path.join(ROOT, "tests", "_emu", "libemu.so")
        srcs = [os.path.join(ROOT, "tests", "emu", "emu.cpp"),
                os.path.join(ROOT, "heavydb_amd", "csrc", "plan.cpp")]
        deps = srcs + [os.path.join(ROOT, "tests", "emu", "emu_atomics.h")] + [os.path.join(ROOT, "heavydb_amd", "csrc", h)
                       for h in ("rowfunc.h", "dev_common.h", "plan.h", "expr.h")] + \
            [os.path.join(ROOT, "include", "mi355q.h")]
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMQ_EMU", "-Wall",
                            "-Wno-unused-function"] + srcs + ["-o", out], check=True)
        l = C.CDLL(out)
        P = C.POINTER
        l.emu_qmd_init.restype = C.c_int32
        l.emu_qmd_init.argtypes = [P(capi.Plan), P(capi.QMD)]
        l.emu_execute.restype = C.c_int32
        l.emu_execute.argtypes = [P(capi.Plan), P(capi.Inputs), C.c_int, C.c_void_p, C.c_int64,
                                  C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, P(capi.QMD)]
        for fn in ("emu_buffer_bytes", "emu_group_col_offset", "emu_slot_col_offset"):
            getattr(l, fn).restype = C.c_int64
        l.emu_buffer_bytes.argtypes = [P(capi.QMD)]
        l.emu_group_col_offset.argtypes = [P(capi.QMD), C.c_int]
        l.emu_slot_col_offset.argtypes = [P(capi.QMD), C.c_int]
        l.emu_init_buffer.restype = None
        l.emu_init_buffer.argtypes = [P(capi.QMD), C.c_void_p]
        l.emu_reduce.restype = C.c_int32
        l.emu_reduce.argtypes = [P(capi.QMD), C.c_void_p, C.c_void_p, C.c_int64]
        _emu = l
    return _emu


def decode_join_table(raw: np.ndarray, hash_type: int, entries: int, kc: int, w: int, min_key: int = 0):
    """{key tuple: sorted row ids} of a hash join buffer — HashTable::toSet() of the reference's
    JoinHashTableTest (slot positions and the order inside a payload run depend on the build
    order, so tables are compared decoded)."""
    out = {}
    if hash_type == 0:
        slots = raw.view(np.int32)[:entries]
        return {(int(min_key + i),): [int(v)] for i, v in enumerate(slots) if v >= 0}
    dt = np.int32 if w == 4 else np.int64
    empty = 2**31 - 1 if w == 4 else 2**63 - 1
    if hash_type == 1:
        tab = raw[:entries * (kc + 1) * w].view(dt).reshape(entries, kc + 1)
        return {tuple(int(x) for x in r[:kc]): [int(r[kc])] for r in tab if r[0] != empty}
    key_bytes = 0 if hash_type == 2 else entries * kc * w
    i32 = raw[key_bytes:].view(np.int32)
    offsets, counts, payloads = i32[:entries], i32[entries:2 * entries], i32[2 * entries:]
    keys = None if hash_type == 2 else raw[:key_bytes].view(dt).reshape(entries, kc)
    for e in range(entries):
        if offsets[e] < 0:
            assert counts[e] == 0
            continue
        k = (int(min_key + e),) if hash_type == 2 else tuple(int(x) for x in keys[e])
        out[k] = sorted(int(x) for x in payloads[offsets[e]:offsets[e] + counts[e]])
    return out


_hostsim = {}


def hostsim_lib(real_fast: bool = False) -> str:
    """tests/hostsim: the library's host code (api.cpp, plan.cpp) and kernels_generic.hip compiled for the CPU against
    a stand-in HIP runtime (memory = host memory poisoned with 0xA5, launches on a pool of host threads), the fast
    kernel families replaced by row-function stand-ins (tests/hostsim/kernels_host.cpp).  Returns the path of the
    built library (same C ABI as libmi355q.so).

    real_fast=True: a second library in which EVERY kernel file is the real device source compiled for the host (blocks
    run as 1024 cooperative fibers): kernels_fast.hip, kernels_lds.hip, kernels_part.hip (the producer / flusher pipeline
    of the scatter works because every polling loop of the device code sleeps, and s_sleep is a fiber yield here; waits
    for OTHER workgroups are bounded on the device and simply expire here, blocks run one after the other) and
    kernels_sort.hip (top-k selection and the hand-written one-sweep radix sort)."""
    key = bool(real_fast)
    if _hostsim.get(key) is not None:
        return _hostsim[key]
    src_dir = os.path.join(ROOT, "tests", "hostsim")
    csrc = os.path.join(ROOT, "heavydb_amd", "csrc")
    san = os.environ.get("MI355Q_HOSTSIM_SANITIZE", "")   # "address": an ASan build of the simulation (run under LD_PRELOAD of the runtime)
    out_dir = os.path.join(ROOT, "tests", ("_hostsim_real" if real_fast else "_hostsim") + ("_" + san if san else ""))
    out = os.path.join(out_dir, "libmi355q_hostsim_real.so" if real_fast else "libmi355q_hostsim.so")
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = "g++"
    real_srcs = ["kernels_fast.hip", "kernels_lds.hip", "kernels_part.hip", "kernels_sort.hip", "kernels_idx.hip", "fast_common.h", "lds_args.h"] if real_fast else []
    deps = [os.path.join(src_dir, f) for f in ("hip_host.cpp", "kernels_host.cpp", "shim/hip/hip_runtime.h",
                                               "shim/hip/hip_runtime_api.h")] + \
        [os.path.join(csrc, f) for f in ["api.cpp", "api_projection.cpp", "api_result.cpp", "api_join.cpp", "api_internal.h", "boolfilter.cpp", "boolfilter.h", "plan.cpp", "kernels_generic.hip",
                                         "kernels_proj.hip", "kernels_filter.hip", "regprog.h", "kernels.h", "rowfunc.h", "dev_common.h", "plan.h", "expr.h",
                                         "fast_common.h"] + real_srcs] + \
        [os.path.join(ROOT, "include", "mi355q.h"), os.path.abspath(__file__)]   # (the build recipe patches the sources)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(out_dir, exist_ok=True)
        import re
        # the one construct the stand-in cannot express: dynamic LDS declared `extern __shared__`
        with open(os.path.join(csrc, "kernels_generic.hip")) as f:
            kg = f.read()
        kg_host, n_sub = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) int64_t (\w+)\[\];",
                                 r"int64_t* const \1 = (int64_t*)hipsim::dynamic_shared();", kg)
        assert n_sub == 2 and "extern __shared__" not in kg_host
        kg_cpp = os.path.join(out_dir, "kernels_generic_host.cpp")
        with open(kg_cpp, "w") as f:
            f.write(kg_host)
        # kernels whose body holds a barrier, a wave shuffle / ballot or LDS run as fibers; the rest as plain loops
        names, plain = [], []
        with open(os.path.join(csrc, "kernels_proj.hip")) as f:
            kp_src = f.read()
        with open(os.path.join(csrc, "kernels_filter.hip")) as f:
            kf_src = f.read()   # (the row-mask pre-pass of compiled filters: the real device source in both simulations, no dynamic LDS)
        for text in (kg, kp_src, kf_src):
            for m in re.finditer(r"__global__[^{;]*?void\s+(k_\w+)\s*\(", text):
                depth, i = 0, text.index("{", m.end())
                start = i
                while True:
                    depth += {"{": 1, "}": -1}.get(text[i], 0)
                    i += 1
                    if depth == 0:
                        break
                (names if re.search(r"__syncthreads|__shfl|__ballot|__shared__|__any", text[start:i]) else plain).append(m.group(1))
        with open(os.path.join(out_dir, "barrier_kernels.inc"), "w") as f:
            f.write("".join(f'    "{n}",\n' for n in names))
        with open(os.path.join(out_dir, "plain_kernels.inc"), "w") as f:
            f.write("".join(f'    "{n}",\n' for n in plain))
        flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-pthread", "-w", "-DHOSTSIM_DEVICE_CODE", "-I" + os.path.join(src_dir, "shim"), "-I" + csrc,
                 "-I" + os.path.join(ROOT, "include"), "-I" + out_dir]
        if san:
            flags += ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-shared-libsan"]
        # the Projection family (kernels_proj.hip) is the real device source in BOTH simulations: its workgroups take tiles
        # off a ticket counter, so a tile's predecessors are always finished when blocks run one after the other
        with open(os.path.join(csrc, "kernels_proj.hip")) as f:
            kp = f.read()
        kp, n_sub = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char (\w+)\[\];",
                            r"char* const \1 = (char*)hipsim::dynamic_shared();", kp)
        assert n_sub == 2 and "extern __shared__" not in kp
        kp_cpp = os.path.join(out_dir, "kernels_proj_host.cpp")
        with open(kp_cpp, "w") as f:
            f.write(kp)
        kf_cpp = os.path.join(out_dir, "kernels_filter_host.cpp")
        with open(kf_cpp, "w") as f:
            f.write(kf_src)
        srcs = [os.path.join(csrc, "api.cpp"), os.path.join(csrc, "api_projection.cpp"), os.path.join(csrc, "api_result.cpp"),
                os.path.join(csrc, "api_join.cpp"), os.path.join(csrc, "boolfilter.cpp"),
                os.path.join(csrc, "plan.cpp"), kg_cpp,
                kp_cpp, kf_cpp, os.path.join(src_dir, "kernels_host.cpp"), os.path.join(src_dir, "hip_host.cpp")]
        if real_fast:
            flags.append("-DHOSTSIM_REAL_FAST")
            for name in ("kernels_fast.hip", "kernels_lds.hip", "kernels_part.hip", "kernels_sort.hip", "kernels_idx.hip"):
                with open(os.path.join(csrc, name)) as f:
                    src = f.read()
                src, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char (\w+)\[\];",
                                 r"char* const \1 = (char*)hipsim::dynamic_shared();", src)
                assert "extern __shared__" not in src, name
                if name == "kernels_part.hip":
                    # the LDS-only barrier is an inline-asm s_barrier
                    pat = 'asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");'
                    assert src.count(pat) == 1
                    src = src.replace(pat, "__syncthreads();")
                    # a wave runs in lockstep on the device; the fibers of a flusher wave meet before the line they
                    # have just read is declared free
                    pat = '      asm volatile("" ::: "memory");\n      if (need) {\n        my_fl[k] += 1;'
                    assert src.count(pat) == 1
                    src = src.replace(pat, "      hipsim::wave_sync();\n      if (need) {\n        my_fl[k] += 1;")
                    # values every lane of a wave reads from LDS "at the same instant" are wave-uniform on the device; here
                    # the fibers read at different times, so lane 0's reading is broadcast where control flow depends on it
                    pat = "const bool all_done = lds_peek(done) == (uint32_t)kProdWaves;"
                    assert src.count(pat) == 1
                    src = src.replace(pat, "const bool all_done = __shfl((int)(lds_peek(done) == (uint32_t)kProdWaves), 0) != 0;")
                    # ... and the lanes that flush the segments of ONE staging line must agree on how far that line has
                    # been written (one LDS read in lockstep on the device): the reading of the line's first lane counts
                    pat = "const uint32_t w = lds_peek(&written[p]);"
                    assert src.count(pat) == 2
                    src = src.replace(pat, "const uint32_t w = (uint32_t)__shfl((int)lds_peek(&written[p]), "
                                           "(int)((threadIdx.x & 63) - (sidx & 63)));")
                    # lanes of one wave wait for the lane that fetches the next spill block: that lane must get to run
                    pat = "if (w == kSpillBusy) continue;"
                    assert src.count(pat) == 1
                    src = src.replace(pat, "if (w == kSpillBusy) { hipsim::fiber_yield(); continue; }")
                if name == "kernels_idx.hip":
                    # the same producer / flusher pipeline as k_part_scatter: the same four stand-ins for wave lockstep
                    pat = '      asm volatile("" ::: "memory");\n      if (need) {\n        my_fl[k] += 1;'
                    assert src.count(pat) == 1
                    src = src.replace(pat, "      hipsim::wave_sync();\n      if (need) {\n        my_fl[k] += 1;")
                    pat = "const bool all_done = idx_peek(done) == (uint32_t)kIdxProdWaves;"
                    assert src.count(pat) == 1
                    src = src.replace(pat, "const bool all_done = __shfl((int)(idx_peek(done) == (uint32_t)kIdxProdWaves), 0) != 0;")
                    pat = "const uint32_t w = idx_peek(&written[p]);"
                    assert src.count(pat) == 2
                    src = src.replace(pat, "const uint32_t w = (uint32_t)__shfl((int)idx_peek(&written[p]), "
                                           "(int)((threadIdx.x & 63) - (sidx & 63)));")
                    pat = "if (w == kIdxSpillBusy) continue;"
                    assert src.count(pat) == 1
                    src = src.replace(pat, "if (w == kIdxSpillBusy) { hipsim::fiber_yield(); continue; }")
                cpp = os.path.join(out_dir, name.replace(".hip", "_host.cpp"))
                with open(cpp, "w") as f:
                    f.write(src)
                srcs.append(cpp)
        objs = []
        for src in srcs:
            obj = os.path.join(out_dir, os.path.basename(src) + ".o")
            subprocess.run([cxx] + flags + ["-c", src, "-o", obj], check=True)
            objs.append(obj)
        subprocess.run([cxx, "-shared", "-pthread", "-Wl,-Bsymbolic", "-o", out] + objs +
                       (["-fsanitize=" + san, "-shared-libsan"] if san else []), check=True)
    _hostsim[key] = out
    return out
