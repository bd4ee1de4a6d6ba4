#!/bin/bash
# development helper (GPU box): GPU tests, default bench, CLI e2e variants, LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-check2}; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1
timeout 900 python bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
with open("/dev/shm/c3.fa", "wb") as f:
    for i in range(100000):
        f.write(b">syn_%d\n" % i + TH.synth_read(lib, 1, i, 1000, 10, 0.10) + b"\n")
PY
for v in "2 65536" "1 65536" "2 32768" "2 100000" "2 16384"; do set -- $v
  echo "== workers/dev $1 batch $2" >> $O/e2e.txt
  TH_WORKERS_PER_DEV=$1 TH_TRACE=1 $R/tidehunter_amd/TideHunter-amd -b $2 -o /dev/shm/c3.out /dev/shm/c3.fa 2>&1 | grep "run_file\|Real time" >> $O/e2e.txt
  echo "records: $(grep -c '^>' /dev/shm/c3.out) md5: $(md5sum < /dev/shm/c3.out)" >> $O/e2e.txt
done
rm -f /dev/shm/c3.fa /dev/shm/c3.out
bash tools/pmc_lds.sh > $O/pmc_lds.log 2>&1
tail -4 $O/pytest_gpu.log; cat $O/e2e.txt; tail -30 $O/pmc_lds.log; python3 -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['value_e2e'])"
