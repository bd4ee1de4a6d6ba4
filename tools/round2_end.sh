#!/bin/bash
# Round-2 measurement recipe (GPU box, through gpurun): writes everything under gpurun_out/round2/, named as it is copied into
# profiles/ (r02_*).  Counters are collected in their own rocprofv3 passes with --kernel-trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round2; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/r02_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $O/r02_pytest_gpu.txt 2>&1
timeout 900 python3 bench.py > $O/r02_bench.json 2> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r02_kernel_stats.csv 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e 0 > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e 0 > $O/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/round2")
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
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0` (config 3); "
                   "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per the gfx950 correction in MI355X_MICROARCH.md",
           "reads_per_launch": 100000 // max(1, max([v.get("launches", 1) for k, v in res.items() if "th_k_poa" in k] or [1])), "config": "c3", "kernels": res}, open(os.path.join(O, "r02_hbm_traffic.json"), "w"), indent=1)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
bash tools/pmc_instr.sh > $O/pmc_instr_summary.txt 2>&1; cp $R/gpurun_out/pmc_instr/pmc_instr.json $O/r02_pmc_instr.json
TH_BENCH_BATCH=100000 bash tools/pmc_lds.sh > $O/pmc_lds_summary.txt 2>&1; cp $R/gpurun_out/pmc_lds/pmc_lds.json $O/r02_pmc_lds.json
( for c in c2 c4 c5; do timeout 600 python bench.py --config $c --cpu-sample 0 --steps 2 --warmup 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$c', 'reads/s', round(j['value']), 'Mbases/s', round(j['mbases_per_s']), 'ms/step', round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()}, 'e2e reads/s', round(j['value_e2e']['value']) if 'value_e2e' in j else None)"; done ) > $O/r02_other_shapes.txt 2>&1
E2E_CASES="2:32768:0 2:32768:0 1:32768:0 2:65536:0" bash tools/gpu_e2e.sh round2 > /dev/null 2>&1; grep -v "^\[th\] " $O/e2e.txt > $O/r02_cli_e2e.txt
cp $O/r02_hbm_traffic.json $R/profiles/r02_hbm_traffic.json; timeout 900 python3 bench.py --cpu-sample 0 --e2e 0 > $O/r02_bench_quoting_traffic.json 2>/dev/null
cat $O/r02_pytest_gpu.txt $O/r02_other_shapes.txt; tail -c 1200 $O/r02_bench.json; cat $O/pmc_instr_summary.txt | tail -12; head -12 $O/r02_kernel_stats.csv | cut -c1-150
