#!/bin/bash
# development helper (GPU box): 1 / 2 / 3 / 4 concurrent parts per batch, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in 2 3 4 1; do for c in c3 c5 c2; do
  echo -n "TH_SUBBATCH=$v $c: "
  TH_SUBBATCH=$v timeout 600 python bench.py --config $c --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()})"
done; done
