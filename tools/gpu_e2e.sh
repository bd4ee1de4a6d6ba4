#!/bin/bash
# development helper (GPU box): CLI e2e with full traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-e2e}; mkdir -p $O; cd $R
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
with open("/dev/shm/c3.fa", "wb") as f:
    for i in range(100000):
        f.write(b">syn_%d\n" % i + TH.synth_read(lib, 1, i, 1000, 10, 0.10) + b"\n")
PY
rm -f $O/e2e.txt
for v in ${E2E_CASES:-1:65536:1000000 2:65536:0 2:32768:0}; do set -- ${v//:/ }
  echo "== workers/dev $1 batch $2 ramp0 $3 sub ${4:-1} submin ${5:-16384}" >> $O/e2e.txt
  if [ "$3" = "0" ]; then unset TH_RAMP0; else export TH_RAMP0=$3; fi
  TH_FILE_SUB=${4:-1} TH_SUBBATCH_MIN=${5:-16384} TH_WORKERS_PER_DEV=$1 TH_TRACE=1 $R/tidehunter_amd/TideHunter-amd -b $2 -o /dev/shm/c3.out /dev/shm/c3.fa 2>&1 | grep -v "^$" >> $O/e2e.txt
  echo "records: $(grep -c '^>' /dev/shm/c3.out) md5: $(md5sum < /dev/shm/c3.out)" >> $O/e2e.txt
done
rm -f /dev/shm/c3.fa /dev/shm/c3.out
cat $O/e2e.txt
