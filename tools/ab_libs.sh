#!/bin/bash
# development helper (GPU box): one bench line (per-kernel ms) per build made with tools/build_variant.sh, in a single gpurun call
#   usage: LIBS="libth_a.so libth_b.so" tools/ab_libs.sh [config] [extra bench flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; C=${1:-c3}; shift
for L in $LIBS; do
  for rep in 1 ${REPS:+2}; do
  TH_BENCH_LIB=$R/tests/cpu_emul/$L timeout 600 python3 bench.py --config $C --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --others "" --streamed 0 "$@" 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L', 'reads/s', round(j['value']), 'ms/step', round(j['ms_per_step'],1), {k.replace('th_k_',''): round(v,1) for k,v in j['roofline']['kernel_ms_per_step'].items()})"
  done
done
