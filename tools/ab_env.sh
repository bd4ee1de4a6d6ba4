#!/bin/bash
# development helper (GPU box): one bench line (per-kernel ms) per environment setting of the working tree's library, in a single gpurun call
#   usage: CASES="TH_X=1 TH_X=0,TH_Y=2 -" tools/ab_env.sh [config]      ("-": no setting; a comma joins several settings of one case)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; C=${1:-c3}; shift
for E in ${CASES:--}; do
  ( [ "$E" != "-" ] && for kv in ${E//,/ }; do export "$kv"; done
    timeout 600 python3 bench.py --config $C --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --others "" --streamed 0 "$@" 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$C', '$E', 'reads/s', round(j['value']), 'ms/step', round(j['ms_per_step'],1), {k.replace('th_k_',''): round(v,1) for k,v in j['roofline']['kernel_ms_per_step'].items()})" )
done
