#!/bin/bash
# LDS counters of one bench step (own rocprofv3 passes, kernel trace only): LDS instructions issued, bank-conflict cycles and
# cycles waves waited on LDS instructions, per kernel -> gpurun_out/pmc_lds/pmc_lds.json (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_lds; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
B=${TH_BENCH_BATCH:-32768}; export TH_BENCH_BATCH=$B
rocprofv3 -L > $O/counters_avail.txt 2>&1
pass() { # tag, counters...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/$tag.log 2>&1
}
pass p1 SQ_INSTS_LDS SQ_INSTS_VALU
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass p3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass p4 SQ_WAVE_CYCLES SQ_BUSY_CYCLES
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/pmc_lds")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float); nd = collections.defaultdict(int)
for fn in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    os.remove(fn)
for fn in glob.glob(os.path.join(O, "p1", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]; dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e9; nd[k] += 1
for fn in glob.glob(os.path.join(O, "**", "*kernel_trace.csv"), recursive=True): os.remove(fn)
out = {}
for k in acc:
    if "th_k" not in k: continue
    c = dict(acc[k]); s = dur.get(k, 0.0)
    d = {"launches": nd.get(k, 0), "seconds_under_counters": s, "counters": c}
    if c.get("SQ_INSTS_LDS") and c.get("SQ_INSTS_VALU"): d["lds_per_valu_instr"] = c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"]
    if c.get("SQ_LDS_BANK_CONFLICT") is not None and c.get("SQ_LDS_IDX_ACTIVE"): d["bank_conflict_cycles_per_lds_active_cycle"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAIT_INST_LDS") is not None and c.get("SQ_WAVE_CYCLES"): d["wait_inst_lds_share_of_wave_cycles"] = c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]
    out[k] = d
json.dump({"note": "rocprofv3 --kernel-trace --pmc <pairs> in four separate passes of `bench.py --steps 1 --warmup 0` (config 3, %s reads per launch); "
                   "counters are summed over all dispatches of a kernel and all SEs/XCDs as rocprofv3 reports them" % os.environ.get("TH_BENCH_BATCH"),
           "kernels": out}, open(os.path.join(O, "pmc_lds.json"), "w"), indent=1)
for k, v in out.items(): print(k[:48], {a: ("%.3g" % b) for a, b in v["counters"].items()}, {a: ("%.3g" % b) for a, b in v.items() if isinstance(b, float)})
PY
grep -i "lds" $O/counters_avail.txt | head -40
