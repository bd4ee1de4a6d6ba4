#!/bin/bash
# development helper (GPU box): one pipeline per handle against two concurrent halves
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in 2 1; do for c in c3 c5 c2 c4; do
  echo -n "TH_SUBBATCH=$v $c: "
  TH_SUBBATCH=$v timeout 600 python bench.py --config $c --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()}, j['roofline']['launches_per_step'])"
done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3; E2E_CASES="2:32768:0 2:32768:0" bash tools/gpu_e2e.sh sub | grep -E "run_file\] 1|Real"
