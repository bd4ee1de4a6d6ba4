#!/bin/bash
# Round-4 measurement recipe (GPU box, through gpurun): everything under gpurun_out/round4/, named as it is copied into profiles/ (r04_*).
# Counters are collected in their own rocprofv3 passes with --kernel-trace only; the profiled program is `python3 bench.py --quick`
# (timed steps only: no child processes under the profiler's preload).   SKIP_TESTS=1 skips the GPU test suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round4; mkdir -p $O; cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" = "0" ]; then
  ( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/r04_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $O/r04_pytest_gpu.txt 2>&1
fi
timeout 1700 python3 bench.py > $O/r04_bench.json 2> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --quick > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats.csv 2>/dev/null
find $O/stats -name "*kernel_trace.csv" -delete
cd $R
bash tools/hbm_traffic.sh round4/traffic c3 > $O/traffic_summary.txt 2>&1; cp $O/traffic/hbm_traffic.json $O/r04_hbm_traffic.json
bash tools/pmc_cache.sh round4/cache c3 > $O/cache_summary.txt 2>&1; cp $O/cache/pmc_cache.json $O/r04_pmc_cache.json
bash tools/pmc_instr.sh > $O/pmc_instr_summary.txt 2>&1; cp $R/gpurun_out/pmc_instr/pmc_instr.json $O/r04_pmc_instr.json
TH_BENCH_BATCH=100000 bash tools/pmc_lds.sh > $O/pmc_lds_summary.txt 2>&1; cp $R/gpurun_out/pmc_lds/pmc_lds.json $O/r04_pmc_lds.json
# the command line end to end with its per-batch timeline and host-CPU table (10 x the headline set): two pipelines per device (default), one, four
PAUSE=6 bash tools/e2e.sh round4 10 "2:32768 2:32768 1:65536 4:16384 2:32768:TH_SPIN_SYNC=1" > /dev/null 2>&1
grep -v "^\[th\] " $O/e2e.txt | grep -v "complete at" > $O/r04_cli_e2e.txt
bash tools/reader_rate.sh 10 > $O/r04_reader_rate.txt 2>&1
bash tools/shape_stats.sh "c3 c4 c5" > $O/r04_shape_launches.txt 2>&1
cp $O/r04_hbm_traffic.json $R/profiles/r04_hbm_traffic.json; timeout 600 python3 bench.py --quick > $O/r04_bench_quoting_traffic.json 2>/dev/null
cat $O/r04_pytest_gpu.txt; tail -c 1500 $O/r04_bench.json; tail -9 $O/traffic_summary.txt; tail -9 $O/pmc_instr_summary.txt; grep "host CPU\|run_file\] [0-9]" $O/r04_cli_e2e.txt
