#!/bin/bash
# development helper (GPU box): wall-clock of the CLI on the headline input, first byte parsed to last byte written
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-100000}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
with open("/tmp/c3.fa", "wb") as f:
    for i in range($N):
        f.write(b">syn_p1000_c10_e0.1_s1_%d\n" % i + TH.synth_read(lib, 1, i, 1000, 10, 0.10) + b"\n")
PY
ls -la /tmp/c3.fa
for b in 65536 32768; do
  echo "== batch $b"
  TH_TRACE=1 $R/tidehunter_amd/TideHunter-amd -b $b -o /tmp/c3.out /tmp/c3.fa 2>&1 | grep "process_batch\|run_file\|Real time"
  echo "records: $(grep -c '^>' /tmp/c3.out)  md5: $(md5sum < /tmp/c3.out)"
done
