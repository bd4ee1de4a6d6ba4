#!/bin/bash
# GPU box: L2 hit / miss and memory-side read requests per kernel (own rocprofv3 passes, kernel trace only), so that "served from the
# caches" is a counter: TCC_HIT_sum / TCC_MISS_sum (L2 hit rate, guide section "L2"), TCC_EA0_RDREQ_sum (what leaves L2 for the
# fabric: Infinity Cache or HBM) and TCC_REQ_sum -> gpurun_out/<dir>/pmc_cache.json.   usage: tools/pmc_cache.sh [outdir=cache] [config=c3]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-cache}; CFG=${2:-c3}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum"; do
  t=$(echo $c | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$t -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick --config $CFG > $O/p_$t.log 2>&1
done
python3 - <<PY
import csv, glob, json, os, collections
O = "$O"
res = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
for fn in glob.glob(os.path.join(O, "p_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] += float(row["Counter_Value"]); launches[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    os.remove(fn)
out = {}
for k, v in res.items():
    if "th_k" not in k and "thpoa" not in k: continue
    d = {c: x / max(1, len(launches[(k, c)])) for c, x in v.items()}
    h, m = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
    d["l2_hit_rate"] = h / (h + m) if h + m > 0 else None
    if d.get("TCC_REQ_sum"): d["memory_side_reads_per_l2_request"] = d.get("TCC_EA0_RDREQ_sum", 0.0) / d["TCC_REQ_sum"]
    out[k] = d
json.dump({"note": "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0 --quick --config $CFG, per launch; l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); "
                   "TCC_EA0_RDREQ_sum = read requests that leave L2 for the fabric (Infinity Cache hits are among them: the guide says they are counted, not excluded)",
           "config": "$CFG", "kernels": out}, open(os.path.join(O, "pmc_cache.json"), "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:8]:
    print("%-44s L2 req %.3e hit rate %s  EA rd %.3e (32B %.3e) EA wr %.3e" % (k[:44], d.get("TCC_REQ_sum", 0), ("%.3f" % d["l2_hit_rate"]) if d["l2_hit_rate"] is not None else "-", d.get("TCC_EA0_RDREQ_sum", 0), d.get("TCC_EA0_RDREQ_32B_sum", 0), d.get("TCC_EA0_WRREQ_sum", 0)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
