#!/bin/bash
# development helper (GPU box): POA phase profile + bench with traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-poa}; mkdir -p $O; cd $R
timeout 600 python tools/poa_phase_prof.py > $O/poa_prof.txt 2>&1
TH_TRACE=1 timeout 900 python bench.py --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
grep "poa done" $O/bench.err | tail -12
cat $O/poa_prof.txt | tail -20
python3 -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['roofline']['kernel_ms_per_launch'])"
