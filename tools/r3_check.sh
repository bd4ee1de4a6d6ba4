#!/bin/bash
# development helper (GPU box): GPU tests, then the bench with one / two concurrent parts, then the command line on 10 x the headline set
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r3b}; mkdir -p $O; cd $R
if [ "${SKIP_TESTS:-0}" = "0" ]; then ( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
for sb in ${SUBS:-1 2}; do
  TH_SUBBATCH=$sb timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 > $O/bench_sub$sb.json 2> $O/bench_sub$sb.err
  python3 -c "import json,sys; j=json.load(open('$O/bench_sub$sb.json')); print('sub $sb reads/s', round(j['value']), 'ms/step', round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()})" || tail -5 $O/bench_sub$sb.err
done
if [ "${SKIP_E2E:-0}" = "0" ]; then bash tools/r3_e2e.sh ${1:-r3b} 10 "${E2E_CASES:-1:32768 1:32768 1:65536 2:32768}" 2>&1 | grep "run_file\] \|Real time\|records\|==\|handle ready"; fi
