#!/bin/bash
# Round-5 measurement recipe (GPU box, through gpurun): everything under gpurun_out/round5/, named as it is copied into profiles/ (r05_*).
# Counters are collected in their own rocprofv3 passes with --kernel-trace only; the profiled program is `python3 bench.py --quick`
# (timed steps only: no child processes under the profiler's preload).   SKIP_TESTS=1 skips the GPU test suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round5; mkdir -p $O; cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" = "0" ]; then
  ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/r05_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $O/r05_pytest_gpu.txt 2>&1
fi
# traffic first: the bench line quotes it and checks its csrc hash against the running build
bash tools/hbm_traffic.sh round5/traffic c3 > $O/traffic_summary.txt 2>&1; cp $O/traffic/hbm_traffic.json $O/r05_hbm_traffic.json; cp $O/r05_hbm_traffic.json $R/profiles/r05_hbm_traffic.json
timeout 2400 python3 bench.py > $O/r05_bench.json 2> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --quick > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats.csv 2>/dev/null
find $O/stats -name "*kernel_trace.csv" -delete
cd $R
bash tools/pmc_cache.sh round5/cache c3 > $O/cache_summary.txt 2>&1; cp $O/cache/pmc_cache.json $O/r05_pmc_cache.json
bash tools/pmc_instr.sh > $O/pmc_instr_summary.txt 2>&1; cp $R/gpurun_out/pmc_instr/pmc_instr.json $O/r05_pmc_instr.json
TH_BENCH_BATCH=100000 bash tools/pmc_lds.sh > $O/pmc_lds_summary.txt 2>&1; cp $R/gpurun_out/pmc_lds/pmc_lds.json $O/r05_pmc_lds.json
PAUSE=6 bash tools/e2e.sh round5 10 "2:65536 2:65536 2:32768 1:65536" > /dev/null 2>&1
grep -v "^\[th\] " $O/e2e.txt | grep -v "complete at" > $O/r05_cli_e2e.txt
bash tools/shape_stats.sh "c3 c4 c5" > $O/r05_shape_launches.txt 2>&1
cat $O/r05_pytest_gpu.txt; tail -c 1800 $O/r05_bench.json; tail -9 $O/traffic_summary.txt; tail -9 $O/pmc_instr_summary.txt; grep "host CPU\|run_file\] [0-9]" $O/r05_cli_e2e.txt
