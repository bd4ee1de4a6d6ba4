#!/bin/bash
# development helper (GPU box): GPU tests, then bench of the given configs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r3e}; mkdir -p $O; cd $R
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --others "${OTHERS:-c2,c4,c5}" --streamed 0 > $O/bench.json 2> $O/bench.err
python3 -c "
import json; j=json.load(open('$O/bench.json'))
print('reads/s', round(j['value']), 'ms/step', round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_step'].items()})
for c,v in j.get('other_shapes',{}).items(): print(c, round(v.get('reads_per_s',0)), {k: round(x,1) for k,x in v.get('kernel_ms_per_step',{}).items()}, v.get('error'))
" || tail -5 $O/bench.err
