#!/bin/bash
# development helper (GPU box): the GPU test suite, one default bench line, and the thread scaling of the CPU oracle
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-check}; mkdir -p $O; cd $R
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us >> $O/host.txt 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)\|Thread\|Socket" >> $O/host.txt; free -g | head -2 >> $O/host.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python3 - <<PY > $O/oracle_scaling.txt 2>&1
import sys, subprocess, re; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
with open("/tmp/c3_4096.fa", "wb") as f:
    for i in range(4096):
        f.write(b">r%d\n" % i + TH.synth_read(lib, 1, i, 1000, 10, 0.10) + b"\n")
with open("/tmp/c3_64.fa", "wb") as f:
    for i in range(64):
        f.write(b">r%d\n" % i + TH.synth_read(lib, 1, i, 1000, 10, 0.10) + b"\n")
for t, fn in ((1, "/tmp/c3_64.fa"), (8, "/tmp/c3_4096.fa"), (16, "/tmp/c3_4096.fa"), (32, "/tmp/c3_4096.fa")):
    err = subprocess.run(["$R/oracle/tho_oracle", "-t", str(t), "-o", "/dev/null", fn], capture_output=True, text=True).stderr
    print(err.strip().splitlines()[-1])
PY
tail -5 $O/pytest_gpu.log; cat $O/oracle_scaling.txt; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
