#!/bin/bash
# GPU box: the command line on N x the 100 000-read headline set, TH_TRACE timeline (incl. the host-CPU table) kept.
#   usage: tools/e2e.sh outdir [copies=10] [cases="workers:batch ..."]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-e2e}; mkdir -p $O; cd $R
COPIES=${2:-10}
D=/dev/shm; [ -d $D ] || D=/tmp
python3 - <<PY
import sys, time; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
t0 = time.time()
reads = [TH.synth_read(lib, 1, i, 1000, 10, 0.10) for i in range(100000)]
print("generated in %.1f s" % (time.time() - t0))
with open("$D/c3x.fa", "wb") as f:
    for c in range($COPIES):
        f.write(b"".join(b">syn_%d\n" % (c * 100000 + i) + s + b"\n" for i, s in enumerate(reads)))
print("written in %.1f s" % (time.time() - t0))
PY
ls -la $D/c3x.fa; free -g | head -2
rm -f $O/e2e.txt
for v in ${3:-2:32768 2:32768 1:32768 2:65536}; do set -- ${v//:/ }
  sleep ${PAUSE:-6}       # the driver clears the previous process's HBM for a few seconds; a new process's first allocation waits for it
  echo "== workers/dev $1 batch $2 ${3:+env $3}" >> $O/e2e.txt
  env ${3:-TH_X=0} TH_WORKERS_PER_DEV=$1 TH_TRACE=1 $R/tidehunter_amd/TideHunter-amd -b $2 -o $D/c3x.out $D/c3x.fa 2>&1 | grep -v "^$" >> $O/e2e.txt
  echo "records: $(grep -c '^>' $D/c3x.out) md5: $(md5sum < $D/c3x.out)" >> $O/e2e.txt
done
rm -f $D/c3x.fa $D/c3x.out
grep "run_file\] \|Real time\|records\|==\|handle ready\|upload [1-9]" $O/e2e.txt
