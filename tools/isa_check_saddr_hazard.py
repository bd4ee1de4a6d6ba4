#!/usr/bin/env python3
"""Build-time check: no inline-asm memory instruction may read a scalar address register within 5 wait states of a VECTOR instruction that
wrote it.  gfx9 needs five wait states between "VALU writes SGPR" (v_readlane, v_readfirstlane, v_cmp into an SGPR pair ...) and "VMEM reads
that SGPR"; the compiler's hazard recogniser inserts them for its own instructions but does not look inside inline asm, and th_hw.h's stores
with a scalar base (TH_STORE_U32_SADDR / TH_STORE_U16_SADDR) are inline asm.  When the register allocator spills the base pointer of such a
store to vector lanes and reloads it with v_readlane right in front of the store, the store goes to a stale address: the GPU memory faults of
rounds 3 and 4 that the CPU emulation could not show.   usage: isa_check_saddr_hazard.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
WAIT = 5
bad = 0
hist = []          # (wait states this instruction provides, set of SGPRs it writes from the VALU, text)
in_asm = False
func = '?'
def sregs(tok):
    m = re.fullmatch(r's\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r's(\d+)', tok)
    return {int(m.group(1))} if m else set()
for ln in lines:
    t = ln.strip()
    if re.match(r'^[A-Za-z_.$][\w.$]*:', ln) and not ln.startswith('.L'):
        func = ln.split(':')[0]; hist = []
    if t.startswith(';;#ASMSTART'): in_asm = True; continue
    if t.startswith(';;#ASMEND'): in_asm = False; continue
    if not t or t.startswith((';', '.')) or t.endswith(':'): continue
    op = t.split()[0]
    args = [a.strip() for a in t[len(op):].split(';')[0].split(',')] if len(t) > len(op) else []
    if in_asm and op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        need = set()
        for a in args: need |= sregs(a.split()[0]) if a else set()
        dist = 0
        for ws, wr, txt in reversed(hist):
            if dist >= WAIT: break
            if wr & need:
                print("%s: `%s` reads s%s %d wait state(s) after `%s`" % (func, t, sorted(wr & need), dist, txt)); bad += 1; break
            dist += ws
    # history: what this instruction writes from the vector unit, and the wait states it stands for
    ws = 1
    if op == 's_nop':
        ws = int(args[0], 0) + 1 if args and args[0] else 1
    wr = set()
    if op.startswith('v_') and args:
        if op.startswith(('v_readlane', 'v_readfirstlane')): wr = sregs(args[0])
        elif op.startswith('v_cmp') and not op.endswith('_e32'): wr = sregs(args[0])
        elif '_co_' in op and len(args) > 1: wr = sregs(args[1])
    hist.append((ws, wr, t))
    if len(hist) > 16: hist.pop(0)
print("inline-asm memory instructions with a scalar address too close to a vector write of it:", bad)
sys.exit(1 if bad else 0)
