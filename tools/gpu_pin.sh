#!/bin/bash
# development helper (GPU box): pinned vs pageable host result buffers, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for v in 0 1; do for c in c3 c5; do
  echo -n "rep $rep TH_NOPIN=$v $c: "
  TH_NOPIN=$v timeout 600 python bench.py --config $c --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()})"
done; done; done
