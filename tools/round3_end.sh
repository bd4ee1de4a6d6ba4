#!/bin/bash
# Round-3 measurement recipe (GPU box, through gpurun): writes everything under gpurun_out/round3/, named as it is copied into
# profiles/ (r03_*).  Counters are collected in their own rocprofv3 passes with --kernel-trace only; the profiled program is
# `python3 bench.py --quick` (timed steps only: no child processes under the profiler's preload).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round3; mkdir -p $O; cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" = "0" ]; then
  ( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/r03_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $O/r03_pytest_gpu.txt 2>&1
fi
timeout 1500 python3 bench.py > $O/r03_bench.json 2> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --quick > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats.csv 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/round3")
res = collections.defaultdict(dict)
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for fn in glob.glob(os.path.join(O, tag, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != ctr: continue
            k = row["Kernel_Name"].split("(")[0]
            acc[k] += float(row["Counter_Value"]); cnt[k].add(row["Dispatch_Id"])
        for k in acc:
            res[k][ctr + "_KB_per_launch"] = acc[k] / max(1, len(cnt[k])); res[k]["launches"] = len(cnt[k])
        os.remove(fn)
for k, v in res.items():
    v["hbm_bytes_per_launch_corrected"] = (2 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0 --quick` (config 3, one stream: one launch of "
                   "every kernel per step); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per the gfx950 correction in MI355X_MICROARCH.md",
           "reads_per_launch": 100000, "config": "c3", "kernels": res}, open(os.path.join(O, "r03_hbm_traffic.json"), "w"), indent=1)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
bash tools/pmc_instr.sh > $O/pmc_instr_summary.txt 2>&1; cp $R/gpurun_out/pmc_instr/pmc_instr.json $O/r03_pmc_instr.json
TH_BENCH_BATCH=100000 bash tools/pmc_lds.sh > $O/pmc_lds_summary.txt 2>&1; cp $R/gpurun_out/pmc_lds/pmc_lds.json $O/r03_pmc_lds.json
# the command line end to end with its per-batch timeline (10 x the headline set), one and two pipelines per device
bash tools/r3_e2e.sh round3 10 "2:32768 2:32768 1:65536 1:65536 3:32768" > /dev/null 2>&1; grep -v "^\[th\] " $O/e2e.txt | grep -v "complete at" > $O/r03_cli_e2e.txt
bash tools/r3_shape_stats.sh "c3 c4 c5" > $O/r03_shape_launches.txt 2>&1
cp $O/r03_hbm_traffic.json $R/profiles/r03_hbm_traffic.json; timeout 600 python3 bench.py --quick > $O/r03_bench_quoting_traffic.json 2>/dev/null
cat $O/r03_pytest_gpu.txt; tail -c 1500 $O/r03_bench.json; tail -12 $O/pmc_instr_summary.txt; head -14 $O/r03_kernel_stats.csv | cut -c1-60,200-300
