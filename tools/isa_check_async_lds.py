#!/usr/bin/env python3
"""Build-time check for th_k_chain_dp's untracked LDS read (TH_LDS_READ64_ASYNC, th_hw.h): between the `ds_read_b64` the inline asm
emits and the `s_waitcnt lgkmcnt(0)` of TH_LDS_ASYNC_TAKE / TH_LDS_ASYNC_WAIT the compiler must not touch the destination register pair
(no copy, no spill, no use): it does not know the value is still in flight, and the hardware does not interlock a vector register an LDS
read has yet to write -- an early reader gets the old contents.
Round 5: the walk follows the CONTROL FLOW -- every path from the read, through branches, fall-throughs and the loop's back edge, up to
the first `s_waitcnt` with lgkmcnt(0) on it.  (The first version scanned its basic block only.  When the wait moved to the top of the trip
loop, a register copy the allocator had placed IN FRONT of the wait -- the tied operand of an asm that was nothing but the wait -- went
unseen, and only the 768-entry ring's GPU test caught the stale reads.)
   usage: tools/isa_check_async_lds.py file.s      (hipcc -S output for gfx950; exit code 1 on a violation)"""
import re
import sys


def parse(lines):
    """instructions of one kernel as a flat list [(text, line number)], label -> index of its first instruction"""
    ins, labels = [], {}
    for ln, l in enumerate(lines):
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((';', '.')) or t.endswith(':'):
            continue
        ins.append((t.split(';')[0].strip(), ln))
    return ins, labels


def successors(ins, labels, i):
    t = ins[i][0]
    op = t.split()[0]
    if op == 's_endpgm' or op.startswith('s_setpc'):
        return []
    if op == 's_branch':
        return [labels[t.split()[1]]]
    if op.startswith('s_cbranch'):
        return [labels[t.split()[1]]] + ([i + 1] if i + 1 < len(ins) else [])
    return [i + 1] if i + 1 < len(ins) else []


def vregs(t):
    regs = set(int(x) for x in re.findall(r'\bv(\d+)\b', t))
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', t):
        regs.update(range(int(a), int(b) + 1))
    return regs


ASYNC = re.compile(r'ds_read_b64 v\[(\d+):(\d+)\], v\d+$')          # the asm form: no offset operand


def check(path):
    txt = open(path).read()
    bad, seen = [], 0
    for m in re.finditer(r'^(_Z13th_k_chain_dp\S*):.*?\.end_amdhsa_kernel', txt, re.S | re.M):
        name = m.group(1)
        ins, labels = parse(m.group(0).split('\n'))
        for i, (t, ln) in enumerate(ins):
            r = ASYNC.match(t)
            if not r:
                continue
            seen += 1
            dst = set(range(int(r.group(1)), int(r.group(2)) + 1))
            todo, done = list(successors(ins, labels, i)), set()
            while todo:
                j = todo.pop()
                if j in done:
                    continue
                done.add(j)
                u = ins[j][0]
                if re.match(r's_waitcnt\b.*lgkmcnt\(0\)', u):
                    continue                                     # this path is safe from here on
                if vregs(u) & dst:
                    if ASYNC.match(u):
                        continue                                 # the next request into the same pair: its own walk checks it (LDS returns in order)
                    bad.append("%s: `%s` (line %d) touches v[%d:%d] while the ds_read_b64 of line %d may be in flight" % (name[:40], u, ins[j][1], min(dst), max(dst), ln))
                    continue
                todo.extend(successors(ins, labels, j))
    return seen, bad


if __name__ == "__main__":
    seen, bad = check(sys.argv[1])
    print("%d asynchronous ds_read_b64 in th_k_chain_dp variants, %d violations" % (seen, len(bad)))
    for b in bad[:40]:
        print("  " + b)
    sys.exit(1 if bad or not seen else 0)
