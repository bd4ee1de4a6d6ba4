#!/bin/bash
# development helper (GPU box): randomized byte-identity sweep of the CLI against the CPU oracle over shapes, error rates,
# N content and option sets that the fixed tests do not cover
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<'PY'
import random, os
rng = random.Random(2024)
def noisy(unit, copies, err, pn, rng):
    out = []
    for b in unit * copies:
        x = rng.random()
        if x < err / 3: out.append(rng.choice("ACGT".replace(b, "")))
        elif x < 2 * err / 3: out.append(b); out.append(rng.choice("ACGT"))
        elif x < err: pass
        else: out.append(b)
        if pn and rng.random() < pn: out[-1:] = ["N"]
    return "".join(out)
cfgs = [("p50", 50, 30, 0.05, 0.0, 300), ("p137_e0", 137, 12, 0.0, 0.0, 200), ("p733_e20", 733, 6, 0.20, 0.0, 200),
        ("p2500", 2500, 4, 0.12, 0.0, 100), ("p3500_wide", 3500, 4, 0.10, 0.0, 80), ("p9000", 9000, 3, 0.10, 0.0, 24),
        ("p400_N", 400, 8, 0.08, 0.01, 200), ("p1000_c2", 1000, 3, 0.10, 0.0, 200), ("mixed", 0, 0, 0.10, 0.002, 300)]
for name, period, copies, err, pn, n in cfgs:
    with open("/tmp/sw_%s.fa" % name, "w") as f:
        for i in range(n):
            p = period or rng.choice([31, 64, 200, 999, 1500, 2100]); c = copies or rng.randint(2, 12)
            unit = "".join(rng.choice("ACGT") for _ in range(p))
            flank5 = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 300))); flank3 = "".join(rng.choice("acgt") for _ in range(rng.randint(0, 300)))
            s = flank5 + noisy(unit, c, err, pn, rng) + unit[:rng.randint(0, p - 1)] + flank3
            f.write(">%s_%d\n%s\n" % (name, i, s))
PY
fail=0
for f in /tmp/sw_*.fa; do
  for opt in "" "-f 2" "-f 4 -c 2" "-k 11 -p 20" "-w 4 -H" "-e 0.1 -l" "-M 3 -X 5 -O 5,30 -E 3,1"; do
    $R/tidehunter_amd/TideHunter-amd $opt -o /tmp/gpu.out $f 2>/tmp/gpu.err; rc1=$?
    $R/oracle/tho_oracle -t $(nproc) $opt -o /tmp/cpu.out $f 2>/dev/null; rc2=$?
    if [ $rc1 -eq 0 ] && [ $rc2 -eq 0 ] && cmp -s /tmp/gpu.out /tmp/cpu.out; then echo "ok   $(basename $f) [$opt] $(grep -c . /tmp/gpu.out) lines"; else echo "FAIL $(basename $f) [$opt] rc $rc1/$rc2: $(tail -1 /tmp/gpu.err)"; fail=1; fi
  done
done
echo "sweep done, fail=$fail"
