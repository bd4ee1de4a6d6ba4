#!/bin/bash
# development helper (GPU box): cold command-line run with the allocation trace, GPU tests, bench (one stream), chain-DP knob A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r3d}; mkdir -p $O; cd $R
bash tools/r3_e2e.sh ${1:-r3d} 10 "1:65536 1:65536" > $O/e2e_call.txt 2>&1; grep "th\] \|run_file\] \|Real time\|records\|==\|handle ready" $O/e2e.txt | grep -v "complete at" | head -60
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --others "c2,c5" > $O/bench.json 2> $O/bench.err
python3 -c "
import json; j=json.load(open('$O/bench.json'))
print('reads/s', round(j['value']), 'ms/step', round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_step'].items()}, 'streamed', round(j['value_streamed']['value']))
for c,v in j.get('other_shapes',{}).items(): print(c, round(v.get('reads_per_s',0)), {k: round(x,1) for k,x in v.get('kernel_ms_per_step',{}).items()})
" || tail -5 $O/bench.err
