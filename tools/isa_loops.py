#!/usr/bin/env python
"""Static look at a kernel's gfx950 assembly when no GPU is at hand: total instructions, and for the largest loops
(backward branches) how many VALU / quarter-rate VALU / SALU / LDS / VMEM instructions their bodies hold.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iheavydb_amd/csrc -S --cuda-device-only -o /tmp/k.s heavydb_amd/csrc/kernels_lds.hip
    python tools/isa_loops.py /tmp/k.s <mangled kernel name>

Static counts are an upper bound of what one iteration executes (uniform branches skip the paths of other types), but
they say whether a loop body fits the instruction cache and how branchy the row code is."""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
# locate function
start = next(i for i,l in enumerate(lines) if l.startswith(kern) and l.rstrip().endswith(':') or (l.startswith(kern+':')))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
ins = []  # (idx_in_ins, text)
for l in body:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'):
        continue
    m = re.match(r'^(\.LBB[0-9_]+):', t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if re.match(r'^[a-z_]+[a-z0-9_]*\b', t) and not t.endswith(':'):
        ins.append(t.split(';')[0].strip())
print('total instructions', len(ins))
loops = []
for i, t in enumerate(ins):
    m = re.match(r'^s_cbranch_\w+\s+(\.LBB[0-9_]+)|^s_branch\s+(\.LBB[0-9_]+)', t)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] <= i:
            loops.append((labels[tgt], i, tgt))
def cls(t):
    op = t.split()[0]
    if op.startswith('v_'): 
        return 'valu_q' if re.match(r'v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32|mul_lo_i32)', op) else 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'vmem'
    return 'other'
for a, b, tgt in sorted(loops, key=lambda x: x[0]-x[1])[:8]:
    c = {}
    for t in ins[a:b+1]:
        c[cls(t)] = c.get(cls(t), 0) + 1
    print(tgt, 'len', b - a + 1, c)
