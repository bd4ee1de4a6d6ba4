#!/usr/bin/env python3
"""Build-time check: no inline-asm memory instruction may read a scalar address register within 5 wait states of a VECTOR instruction that
wrote it.  gfx9 needs five wait states between "VALU writes SGPR" (v_readlane, v_readfirstlane, v_cmp into an SGPR pair ...) and "VMEM reads
that SGPR"; the compiler's hazard recogniser inserts them for its own instructions but does not look inside inline asm, and th_hw.h's stores
with a scalar base (TH_STORE_U32_SADDR / TH_STORE_U16_SADDR) are inline asm.  When the register allocator spills the base pointer of such a
store to vector lanes and reloads it with v_readlane right in front of the store, the store goes to a stale address: the GPU memory faults of
rounds 3 and 4 that the CPU emulation could not show.

Round 5: the walk back from a store follows the CONTROL FLOW, not the text -- when a label lies inside the five-wait-state window, the
instructions in front of every branch to that label are searched too (with the wait states left), and the text in front of the label only
if the instruction there can fall through.  csrc/Makefile runs this on the product build's own assembly and fails the build on a finding.
   usage: isa_check_saddr_hazard.py file.s [--all]     (--all: every vector-memory instruction with a scalar operand, not only inline asm)"""
import re, sys
WAIT = 5
check_all = '--all' in sys.argv[2:]


def sregs(tok):
    tok = tok.strip()
    m = re.fullmatch(r's\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r's(\d+)', tok)
    if m: return {int(m.group(1))}
    return {106, 107} if tok == 'vcc' else {106} if tok == 'vcc_lo' else {107} if tok == 'vcc_hi' else set()


class Func:
    def __init__(self, name):
        self.name, self.ins, self.labels_at, self.label_pos, self.branches = name, [], {}, {}, {}

    def finish(self):
        for i, (op, args, *_r) in enumerate(self.ins):
            if op.startswith(('s_cbranch', 's_branch')) and args:
                self.branches.setdefault(args[0], []).append(i)


def parse(path):
    funcs, cur, in_asm, pending = [], None, False, []
    for ln in open(path).read().split('\n'):
        t = ln.strip()
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', ln)
        if m:
            if ln.startswith('.L'):
                if cur is not None: pending.append(m.group(1))
            else:
                if cur is not None: cur.finish()
                cur = Func(m.group(1)); funcs.append(cur); pending = []
            continue
        if t.startswith(';;#ASMSTART'): in_asm = True; continue
        if t.startswith(';;#ASMEND'): in_asm = False; continue
        if cur is None or not t or t.startswith((';', '.')): continue
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(';')[0].split(',')] if len(t) > len(op) else []
        ws = (int(args[0], 0) + 1 if args and args[0] else 1) if op == 's_nop' else 1
        wr = set()
        if op.startswith('v_') and args:
            if op.startswith(('v_readlane', 'v_readfirstlane')): wr = sregs(args[0])
            elif op.startswith('v_cmp'): wr = sregs(args[0]) if not op.endswith('_e32') else {106, 107}
            elif '_co_' in op and len(args) > 1: wr = sregs(args[1])
        idx = len(cur.ins)
        for l in pending: cur.label_pos[l] = idx
        if pending: cur.labels_at[idx] = list(pending); pending = []
        cur.ins.append((op, args, ws, wr, t, in_asm))
    if cur is not None: cur.finish()
    return funcs


def search(f, idx, need, left, seen, via_branch=False):
    """is there a VALU write of a register in `need` fewer than `left` wait states in front of instruction idx (exclusive)?  returns its text or None"""
    while left > 0:
        key = (idx, left, via_branch)
        if key in seen: return None
        seen.add(key)
        if not via_branch:
            for l in f.labels_at.get(idx, ()):      # someone may jump here: the instructions in front of each such branch, the branch itself included
                for b in f.branches.get(l, ()):
                    r = search(f, b + 1, need, left, seen, True)
                    if r: return r
        if idx == 0: return None
        op, args, ws, wr, t, _ = f.ins[idx - 1]
        if not via_branch and op in ('s_branch', 's_endpgm', 's_setpc_b64', 's_swappc_b64'): return None      # nothing falls through from there
        via_branch = False
        if wr & need: return t
        left -= ws; idx -= 1
    return None


bad = n_checked = 0
for f in parse(sys.argv[1]):
    for i, (op, args, ws, wr, t, in_asm) in enumerate(f.ins):
        if not op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) or not (in_asm or check_all): continue
        need = set()
        for a in args: need |= sregs(a.split()[0]) if a else set()
        if not need: continue
        n_checked += 1
        r = search(f, i, need, WAIT, set())
        if r:
            print("%s: `%s` reads a scalar register fewer than %d wait states after `%s`" % (f.name, t, WAIT, r)); bad += 1
print("inline-asm memory instructions with a scalar address too close to a vector write of it: %d (of %d checked)" % (bad, n_checked))
sys.exit(1 if bad else 0)
