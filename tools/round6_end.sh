#!/bin/bash
# Round-6 measurement recipe (GPU box, through gpurun): everything under gpurun_out/round6/, named as it is copied into profiles/ (r06_*).
# Counters are collected in their own rocprofv3 passes with --kernel-trace only; the profiled program is `python3 bench.py --quick`
# (timed steps only: no child processes under the profiler's preload).   SKIP_TESTS=1 skips the GPU test suite, SKIP_HOST=1 the host-ceiling runs.
# Order matters: the traffic, instruction and DP-cell profiles are taken FIRST and copied into profiles/, so that the bench line quotes them
# (and checks the kernel-source + flag hash they were taken with against the running build).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round6; mkdir -p $O; cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" = "0" ]; then
  ( timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/r06_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_pytest_gpu.txt 2>&1
fi
bash tools/hbm_traffic.sh round6/traffic c3 > $O/traffic_summary.txt 2>&1; cp $O/traffic/hbm_traffic.json $O/r06_hbm_traffic.json; cp $O/r06_hbm_traffic.json $R/profiles/r06_hbm_traffic.json
bash tools/pmc_instr.sh > $O/pmc_instr_summary.txt 2>&1; cp $R/gpurun_out/pmc_instr/pmc_instr.json $O/r06_pmc_instr.json; cp $O/r06_pmc_instr.json $R/profiles/r06_pmc_instr.json
# DP cells + the alignment launch by class of item: ONE step of the profiling build (built in the build container: tools/build_variant.sh prof -DTH_ALN_PROF=1 -DTH_POA_PROF=1 -DTH_CHAIN_PROF=1)
if [ -f $R/tests/cpu_emul/libth_prof.so ]; then
  TH_BENCH_LIB=$R/tests/cpu_emul/libth_prof.so timeout 900 python3 bench.py --quick --steps 1 --warmup 0 > $O/prof_c3.json 2> $O/prof_c3.err
  python3 tools/prof_report.py $O/prof_c3.err 100000 c3 $O/r06 > $O/prof_report.txt 2>&1
  cp $O/r06_dp_cells.json $O/r06_align_phases.json $R/profiles/ 2>/dev/null
  grep "poa_prof\|chain_prof" $O/prof_c3.err | tail -4 > $O/r06_poa_chain_events.txt
fi
timeout 2400 python3 bench.py > $O/r06_bench.json 2> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --quick > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats.csv 2>/dev/null
find $O/stats -name "*kernel_trace.csv" -delete
cd $R
bash tools/pmc_cache.sh round6/cache c3 > $O/cache_summary.txt 2>&1; cp $O/cache/pmc_cache.json $O/r06_pmc_cache.json
TH_BENCH_BATCH=100000 bash tools/pmc_lds.sh > $O/pmc_lds_summary.txt 2>&1; cp $R/gpurun_out/pmc_lds/pmc_lds.json $O/r06_pmc_lds.json
PAUSE=6 bash tools/e2e.sh round6 10 "1:65536 1:65536 2:32768" > /dev/null 2>&1
grep -v "^\[th\] " $O/e2e.txt | grep -v "complete at" > $O/r06_cli_e2e.txt
bash tools/shape_stats.sh "c3 c4 c5" > $O/r06_shape_launches.txt 2>&1
if [ "${SKIP_HOST:-0}" = "0" ]; then
  REPLAY_ENV="TH_WAVES_POA=768 TH_WAVES_ALN=1024 TH_WAVES_PART=1024" bash tools/host_ceiling.sh $O/host_ceiling_full.txt "8 4 2 1" > /dev/null 2>&1
  grep -v "complete at\|handle ready\|buffers reserved" $O/host_ceiling_full.txt > $O/r06_host_ceiling.txt
fi
cat $O/r06_pytest_gpu.txt; tail -c 2400 $O/r06_bench.json; tail -9 $O/traffic_summary.txt; tail -9 $O/pmc_instr_summary.txt; grep "host CPU\|run_file\] [0-9]" $O/r06_cli_e2e.txt; grep "==\|steady\|md5" $O/r06_host_ceiling.txt
