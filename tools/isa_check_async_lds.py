#!/usr/bin/env python3
"""Build-time check for th_k_chain_dp's untracked LDS read (TH_LDS_READ64_ASYNC, th_hw.h): between the `ds_read_b64` the inline asm
emits and the `s_waitcnt lgkmcnt(0)` of TH_LDS_ASYNC_WAIT the compiler must not touch the destination register pair (no copy, no
spill, no use): it does not know the value is still in flight, so a different register allocation after a compiler bump could read
stale registers and only the runtime parity tests would notice.  (Checked per basic block.)
   usage: tools/isa_check_async_lds.py file.s      (hipcc -S output for gfx950; exit code 1 on a violation)"""
import re
import sys


def check(path):
    txt = open(path).read()
    bad, seen = [], 0
    for m in re.finditer(r'^(_Z13th_k_chain_dp\S*):.*?\.end_amdhsa_kernel', txt, re.S | re.M):
        name, lines = m.group(1), m.group(0).split('\n')
        pending = None                      # (lo, hi, line number) of an asm ds_read_b64 in flight
        for ln, l in enumerate(lines):
            t = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', t) or t.startswith(('s_branch', 's_cbranch')):
                pending = None              # (the scan is per basic block: the two sources of a trip's records sit in different blocks)
                continue
            if not t or t.startswith((';', '.')) or t.endswith(':'):
                continue
            r = re.match(r'ds_read_b64 v\[(\d+):(\d+)\], v\d+$', t)          # the asm form: no offset operand
            if pending and re.match(r's_waitcnt .*lgkmcnt\(0\)', t):
                pending = None
                continue
            if pending:
                regs = set(int(x) for x in re.findall(r'\bv(\d+)\b', t))
                for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
                    regs.update(range(int(a), int(b) + 1))
                if regs & set(range(pending[0], pending[1] + 1)) and not r:
                    bad.append("%s: `%s` touches v[%d:%d] while its ds_read_b64 (line %d) is in flight" % (name[:40], t, pending[0], pending[1], pending[2]))
            if r:
                pending = (int(r.group(1)), int(r.group(2)), ln)
                seen += 1
    return seen, bad


if __name__ == "__main__":
    seen, bad = check(sys.argv[1])
    print("%d asynchronous ds_read_b64 in th_k_chain_dp variants, %d violations" % (seen, len(bad)))
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad or not seen else 0)
