#!/bin/bash
# development helper (GPU box): bench lines for alternative builds of the library (TH_BENCH_LIB)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in "" $LIBS; do
  if [ -z "$lib" ]; then unset TH_BENCH_LIB; else export TH_BENCH_LIB=$R/tests/cpu_emul/$lib; fi
  echo -n "${lib:-default}: "
  timeout 600 python bench.py --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()})"
done
