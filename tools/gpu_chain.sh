#!/bin/bash
# development helper (GPU box): chain DP with and without the LDS ring
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-chain}; mkdir -p $O; cd $R
for v in 768 512; do
  TH_CHAIN_RING=$v timeout 600 python bench.py --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 > $O/bench_ring$v.json 2> $O/err$v.txt
  python3 -c "import json; j=json.load(open('$O/bench_ring$v.json')); print('ring=$v', round(j['value']), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()})"
done

