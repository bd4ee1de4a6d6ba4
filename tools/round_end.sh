#!/bin/bash
# Round-end recipe (GPU box, through gpurun): GPU tests, smoke, the profile set of tools/prof_round.sh and tools/pmc_instr.sh,
# a second bench line that quotes the fresh traffic profile, the other BASELINE shapes, parity at scale, the VALU issue-rate
# microbenchmark, and the digest of the command line's output on the 100k-read headline file under both alignment engines.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round; mkdir -p $O $R/gpurun_out/pmc_instr
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/prof_round.sh > $O/prof_round.log 2>&1
cp $O/hbm_traffic.json $R/profiles/r02_hbm_traffic.json
timeout 600 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
bash tools/pmc_instr.sh > $R/gpurun_out/pmc_instr/summary.txt 2>&1
( timeout 300 python tools/step_times.py 65536 200 20 0.10 | tail -1; timeout 300 python tools/step_times.py 8192 5000 4 0.15 | tail -1 ) > $O/other_shapes.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/valu_rate tools/valu_rate.hip && timeout 120 /tmp/valu_rate > $O/valu_rate.txt 2>&1
bash tools/parity_scale.sh > $O/parity_scale.txt 2>&1
bash tools/cli_e2e.sh > $O/cli_e2e.txt 2>&1
TH_ALN_MODE=0 $R/tidehunter_amd/TideHunter-amd -b 32768 -o /tmp/c3_strip.out /tmp/c3.fa 2>/dev/null; echo "strip engine only: records $(grep -c '^>' /tmp/c3_strip.out) md5 $(md5sum < /tmp/c3_strip.out)" >> $O/cli_e2e.txt
cat $O/pytest_gpu.log $O/smoke.log $O/other_shapes.txt $O/parity_scale.txt $O/cli_e2e.txt; tail -c 1500 $O/bench_final.json
