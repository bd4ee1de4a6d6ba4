#!/bin/bash
# Measurement recipe for profiles/ (run on the GPU box through gpurun): the bench line, the rocprofv3 kernel summary of
# the same command, and the two HBM counter passes.  Counters are collected in their own runs with --kernel-trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 2 --warmup 1 --quick > $O/stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_write.log 2>&1
# reduce the per-dispatch counter rows on the box (the raw csv is large)
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/round")
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
           "reads_per_launch": int(os.environ.get("TH_BENCH_BATCH", "100000")), "config": "c3", "kernels": res},
          open(os.path.join(O, "hbm_traffic.json"), "w"), indent=1)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -R $O | head -40
