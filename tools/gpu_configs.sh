#!/bin/bash
# development helper (GPU box): one bench line per config shape (c2, c4, c5) -- parity-test shapes, not the headline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-configs}; mkdir -p $O; cd $R
for c in ${CONFIGS:-c2 c4 c5}; do
  TH_TRACE=1 timeout 900 python bench.py --config $c --cpu-sample 0 --e2e ${E2E:-0} --steps 2 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err
  python3 -c "import json; j=json.load(open('$O/bench_$c.json')); print('$c', round(j['value']), round(j['mbases_per_s']), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()}, j.get('value_e2e',{}).get('value'))"
  grep "poa done\|collect" $O/bench_$c.err | tail -6
done
