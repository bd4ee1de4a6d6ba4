#!/bin/bash
# development helper (GPU box): byte-identity sweep with adapters (-5/-3, -F, -s), -r coverage filters, -u, gz and FASTQ input
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<'PY'
import random, gzip
rng = random.Random(77)
rc = lambda s: s.translate(str.maketrans("ACGT", "TGCA"))[::-1]
rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
five, three = rnd(84), rnd(25)
open("/tmp/sw5.fa", "w").write(">5p\n%s\n" % five); open("/tmp/sw3.fa", "w").write(">3p\n%s\n" % three)
def noisy(s, err):
    out = []
    for b in s:
        x = rng.random()
        if x < err / 3: out.append(rng.choice("ACGT".replace(b, "")))
        elif x < 2 * err / 3: out.append(b); out.append(rng.choice("ACGT"))
        elif x < err: pass
        else: out.append(b)
    return "".join(out)
recs = []
for i in range(400):
    p = rng.choice([200, 350, 600, 1200, 2000]); unit = five + rnd(p - 109) + rc(three)
    c = max(2, 4000 // p) if i % 7 else 1          # some single-copy reads for -s
    s = rnd(rng.randint(0, 60)) + noisy(unit * c, rng.choice([0.03, 0.1, 0.15])) + rnd(rng.randint(0, 60))
    recs.append(("ad_%d" % i, rc(s) if i & 1 else s))
with open("/tmp/sw_ad.fa", "w") as f:
    for n, s in recs: f.write(">%s\n%s\n" % (n, s))
with gzip.open("/tmp/sw_ad.fq.gz", "wt") as f:
    for n, s in recs: f.write("@%s some comment\n%s\n+\n%s\n" % (n, s, "".join(rng.choice("!#5AIJ") for _ in s)))
PY
fail=0
chk() { # file, options
  $R/tidehunter_amd/TideHunter-amd $2 -o /tmp/gpu.out $1 2>/tmp/gpu.err; rc1=$?
  $R/oracle/tho_oracle -t $(nproc) $2 -o /tmp/cpu.out $1 2>/dev/null; rc2=$?
  if [ $rc1 -eq 0 ] && [ $rc2 -eq 0 ] && cmp -s /tmp/gpu.out /tmp/cpu.out; then echo "ok   $(basename $1) [$2] $(grep -c . /tmp/gpu.out) lines"; else echo "FAIL $(basename $1) [$2] rc $rc1/$rc2: $(tail -1 /tmp/gpu.err)"; fail=1; fi
}
A="-5 /tmp/sw5.fa -3 /tmp/sw3.fa"
for f in /tmp/sw_ad.fa /tmp/sw_ad.fq.gz; do
  chk $f ""; chk $f "$A"; chk $f "$A -F"; chk $f "$A -F -s"; chk $f "$A -f 2 -a 0.7"; chk $f "$A -f 4 -r 0.5"; chk $f "-r 3 -f 3"; chk $f "-u"; chk $f "-u -f 2 -m 300"; chk $f "$A -F -l -c 3"
done
echo "sweep2 done, fail=$fail"
