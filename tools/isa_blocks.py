#!/usr/bin/env python3
"""development helper: basic blocks of one kernel in a hipcc -save-temps .s file with their VALU / SALU / lane-op / memory counts.
   usage: tools/isa_blocks.py file.s kernel_symbol_prefix [min_instructions]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = next(i for i, l in enumerate(txt) if l.startswith(pref) and l.rstrip().endswith(('@' + l.split(':')[0],)) or (l.startswith(pref) and ':' in l))
end = next(i for i in range(start, len(txt)) if '.end_amdhsa_kernel' in txt[i] or txt[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in txt[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1); continue
    t = l.strip()
    if not t or t.startswith((';', '.')): continue
    cur.append(t)
blocks.append((name, cur))
tot = 0
for name, ins in blocks:
    ops = [i.split()[0] for i in ins]
    v = sum(o.startswith('v_') for o in ops); s = sum(o.startswith('s_') for o in ops)
    lane = sum(o.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')) for o in ops)
    dpp = sum('dpp' in i or 'row_' in i or 'wave_' in i for i in ins)
    mem = sum(o.startswith(('global_', 'flat_', 'scratch_', 'buffer_')) for o in ops); lds = sum(o.startswith('ds_') for o in ops)
    wait = sum(o.startswith('s_waitcnt') for o in ops); nop = sum(o.startswith('s_nop') for o in ops)
    pk = sum(o.startswith('v_pk_') for o in ops)
    br = [i for i in ins if i.startswith(('s_cbranch', 's_branch'))]
    if len(ins) >= mn:
        print("%-12s n %4d  VALU %4d (lane %3d dpp %3d pk %3d)  SALU %4d (wait %2d nop %2d)  mem %3d lds %3d  -> %s" % (name, len(ins), v, lane, dpp, pk, s, wait, nop, mem, lds, ' '.join(b.split()[-1] for b in br)))
