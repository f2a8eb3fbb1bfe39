#!/usr/bin/env python3
"""Developer tool: instruction mix of one kernel in a hipcc -S listing, whole kernel and per
basic block (label to label), largest blocks first.
   hipcc --offload-arch=gfx950 -O3 ... --offload-device-only -S x.hip -o x.s
   python tools/isa_count.py x.s <kernel-name-substring> [blocks]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^[A-Za-z_][\w$.]*:', l) and key in l and not l.startswith('.L'):
        start = i
        break
if start is None:
    sys.exit('kernel not found')
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
ins = re.compile(r'^\s+([a-z][a-z0-9_]+)\s')
blocks, cur, name = [], collections.Counter(), 'entry'
for l in lines[start + 1:end + 1]:
    m = re.match(r'^(\.LBB[\w]+):', l)
    if m:
        blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1)
        continue
    m = ins.match(l)
    if m and not m.group(1).startswith('.'):
        cur[m.group(1)] += 1
blocks.append((name, cur))
tot = collections.Counter()
for _, c in blocks:
    tot.update(c)
def cls(c):
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_mfma'): g['mfma'] += v
        elif k.startswith('v_pk_'): g['v_pk'] += v
        elif k in ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_rsq_f32', 'v_sqrt_f32'): g['trans'] += v
        elif k.startswith('v_'): g['valu'] += v
        elif k.startswith('ds_'): g['lds'] += v
        elif k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): g['vmem'] += v
        elif k.startswith('s_'): g['salu'] += v
    return dict(g)
print(lines[start][:100])
print('whole kernel:', sum(tot.values()), cls(tot))
for name, c in sorted(blocks, key=lambda b: -sum(b[1].values()))[:nblocks]:
    print(f'-- block {name}: {sum(c.values())} instructions', cls(c))
    print('   ', dict(c.most_common(40)))
