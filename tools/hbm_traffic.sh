#!/bin/bash
# GPU box: HBM traffic per kernel launch from the L2's memory-side counters, FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (--kernel-trace --pmc only) of `python3 bench.py --steps 1 --warmup 0 --quick` -> gpurun_out/<dir>/hbm_traffic.json.
# Both bounds of the gfx950 FETCH correction are reported: x2 (the guide's calibration for 16-byte-per-lane streaming reads) and x1
# (narrow, scattered reads -- POA tiles, chain-DP records -- are uncalibrated and lie between the two).
#   usage: tools/hbm_traffic.sh [outdir=traffic] [config=c3]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-traffic}; CFG=${2:-c3}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick --config $CFG > $O/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, os, collections, sys
sys.path.insert(0, "$R")
O = "$O"
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(os.path.join(O, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != ctr: continue
            k = row["Kernel_Name"].split("(")[0]
            acc[k] += float(row["Counter_Value"]); cnt[k].add(row["Dispatch_Id"])
        for k in acc:
            res[k][ctr + "_KB_per_launch"] = acc[k] / max(1, len(cnt[k])); res[k]["launches"] = len(cnt[k])
        os.remove(fn)
for k, v in res.items():
    f, w = v.get("FETCH_SIZE_KB_per_launch", 0.0), v.get("WRITE_SIZE_KB_per_launch", 0.0)
    v["hbm_bytes_per_launch_corrected"] = (2 * f + w) * 1024           # the guide's gfx950 correction (upper bound for narrow reads)
    v["hbm_bytes_per_launch_uncorrected"] = (f + w) * 1024             # lower bound
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --steps 1 --warmup 0 --quick (one stream: one launch of every kernel "
                   "per step); corrected = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (calibrated for 16-byte-per-lane streaming reads), uncorrected = "
                   "(FETCH_SIZE + WRITE_SIZE) * 1024: narrow scattered reads lie between the two",
           "reads_per_launch": {"c3": 100000, "c2": 65536, "c4": 20000, "c5": 100000}["$CFG"], "config": "$CFG", "csrc_hash": __import__("bench").csrc_hash(), "kernels": res}, open(os.path.join(O, "hbm_traffic.json"), "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"])[:8]:
    print("%-60s fetch %8.1f GB  write %8.1f GB  corrected %8.1f GB" % (k[:60], v.get("FETCH_SIZE_KB_per_launch", 0) / 1e6 * 1.024, v.get("WRITE_SIZE_KB_per_launch", 0) / 1e6 * 1.024, v["hbm_bytes_per_launch_corrected"] / 1e9))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
