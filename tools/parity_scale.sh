#!/bin/bash
# development helper (GPU box): byte-identity of the CLI against the CPU oracle on thousands of synthetic reads per shape
R=${GRAFT_REPO_ROOT:-/root/repo}
gen() { python3 - "$@" <<'PY'
import sys; sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tidehunter_amd as TH
lib = TH.load_library()
out, n, period, copies, err, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
with open(out, "wb") as f:
    for i in range(n):
        f.write(b">r%d\n" % i + TH.synth_read(lib, seed, i, period, copies, err) + b"\n")
PY
}
check() { # name, file, extra options
  $R/tidehunter_amd/TideHunter-amd $3 -o /tmp/gpu.out $2 2>/dev/null
  $R/oracle/tho_oracle -t $(nproc) $3 -o /tmp/cpu.out $2 2>/dev/null
  if cmp -s /tmp/gpu.out /tmp/cpu.out; then echo "$1 [$3]: identical ($(grep -c . /tmp/gpu.out) lines, md5 $(md5sum < /tmp/gpu.out | cut -c1-12))"; else echo "$1 [$3]: DIFFERENT"; fi
}
gen /tmp/c3.fa 4096 1000 10 0.10 1; check C3 /tmp/c3.fa ""; check C3 /tmp/c3.fa "-f 2"; check C3 /tmp/c3.fa "-f 4 -c 3"
gen /tmp/c2.fa 4096 200 20 0.10 2;  check C2 /tmp/c2.fa ""; check C2 /tmp/c2.fa "-u -f 2"; check C2 /tmp/c2.fa "-k 10 -w 5"; check C2 /tmp/c2.fa "-H"
gen /tmp/c4.fa 512 5000 4 0.15 3;   check C4 /tmp/c4.fa ""; check C4 /tmp/c4.fa "-f 2 -l"
