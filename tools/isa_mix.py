#!/usr/bin/env python3
"""instruction mix per basic block of one kernel in a gfx950 .s file:  tools/isa_mix.py file.s <mangled-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(sys.argv[2]) + r'\w*:', l))
bb = collections.OrderedDict(); cur = 'entry'; bb[cur] = []
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: cur = m.group(1); bb[cur] = []; continue
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'): continue
    bb[cur].append(l)
tot = collections.Counter()
for k, v in bb.items():
    c = collections.Counter()
    for i in v:
        op = i.split()[0]
        key = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
               'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'scratch' if op.startswith('scratch_') else
               'wait' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop') else 'smem' if op.startswith(('s_load', 's_buffer')) else
               'branch' if op.startswith(('s_cbranch', 's_branch')) else 'salu' if op.startswith('s_') else 'other')
        c[key] += 1
    tot.update(c)
    if len(v) >= 8: print(f"{k:12s} {len(v):5d}", dict(c))
print("total", dict(tot))
