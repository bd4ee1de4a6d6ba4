#!/bin/bash
# development helper (GPU box): GPU tests, default bench, CLI e2e with per-batch traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-check3}; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest_gpu.log 2>&1
timeout 900 python bench.py --cpu-sample 0 --e2e 0 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
with open("/dev/shm/c3.fa", "wb") as f:
    for i in range(100000):
        f.write(b">syn_%d\n" % i + TH.synth_read(lib, 1, i, 1000, 10, 0.10) + b"\n")
PY
for v in "2 65536 0" "1 65536 0" "2 65536 1000000" "1 65536 1000000" "2 32768 0" "2 32768 1000000"; do set -- $v
  echo "== workers/dev $1 batch $2 ramp0 $3" >> $O/e2e.txt
  if [ "$3" = "0" ]; then unset TH_RAMP0; else export TH_RAMP0=$3; fi
  TH_WORKERS_PER_DEV=$1 TH_TRACE=1 $R/tidehunter_amd/TideHunter-amd -b $2 -o /dev/shm/c3.out /dev/shm/c3.fa 2>&1 | grep "run_file\|Real time\|process_batch" >> $O/e2e.txt
  echo "records: $(grep -c '^>' /dev/shm/c3.out) md5: $(md5sum < /dev/shm/c3.out)" >> $O/e2e.txt
done
rm -f /dev/shm/c3.fa /dev/shm/c3.out
tail -4 $O/pytest_gpu.log; cat $O/e2e.txt; python3 -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['roofline']['kernel_ms_per_launch'])"
