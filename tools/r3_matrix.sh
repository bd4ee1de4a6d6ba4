#!/bin/bash
# development helper (GPU box): cold command-line run first, then a matrix of pipelines x batch sizes, then the bench and a kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r3c}; mkdir -p $O; cd $R
bash tools/r3_e2e.sh ${1:-r3c} 10 "${E2E_CASES:-1:65536 2:65536 1:131072 2:32768 1:65536 1:32768:TH_FILE_SUB=2}" 2>&1 | grep "run_file\] \|Real time\|records\|==\|handle ready\|generated\|written"
if [ "${SKIP_BENCH:-0}" = "0" ]; then
  timeout 1500 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json
  cd /tmp; export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --others "" --streamed 0 > $O/stats.log 2>&1
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; find $O/stats -name "*.csv" -size +1M -delete
  head -25 $O/kernel_stats.csv | cut -c1-60,200-330
fi
