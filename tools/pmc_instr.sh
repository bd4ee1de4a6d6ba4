#!/bin/bash
# Instruction counters of one bench step (own rocprofv3 pass, kernel trace only): VALU / SALU wave-instructions per launch and how busy they keep the two issue
# ports.  Prices from profiles/r05_valu_rate.txt (tools/valu_rate.hip, measured on this chip at 2.39 GHz): a vector instruction of the class these kernels are
# made of (max, compare, v_pk_*, DPP, VOP3, lane ops) issues every 4.2 cycles per SIMD, the fast class (v_add/sub/and/or/xor/mov with vector operands) every 2.5;
# a scalar instruction every 4.25 cycles per SIMD.  valu_busy_upper prices every vector instruction at 4.2 (an upper bound: the fast class is not counted apart),
# salu_busy prices s_nop like any other (also an upper bound: s_nop issues in one cycle).  Round 4 called a flat 4-cycle price "the fraction of slots".
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_instr; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick $BENCH_ARGS > $O/log 2>&1
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/pmc_instr")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
for fn in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    os.remove(fn)
for fn in glob.glob(os.path.join(O, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        dur[row["Kernel_Name"].split("(")[0]] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e9
    os.remove(fn)
out = {}
for k in acc:
    if "th_k" not in k: continue
    valu, salu, s = acc[k].get("SQ_INSTS_VALU", 0.0), acc[k].get("SQ_INSTS_SALU", 0.0), dur.get(k, 0.0)
    out[k] = {"launches": len(n[k]), "valu_wave_instr": valu, "salu_wave_instr": salu, "seconds_under_counters": s,
              "valu_busy_upper": (valu * 4.2 / (1024 * 2.39e9 * s)) if s > 0 else None, "salu_busy_upper": (salu * 4.25 / (1024 * 2.39e9 * s)) if s > 0 else None}
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); import bench
json.dump({"csrc_hash": bench.csrc_hash(), "config": "c3", "reads_per_launch": int(os.environ.get("TH_BENCH_BATCH", "100000")), "note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU, one pass of `bench.py --steps 1 --warmup 0` (%d reads of config 3 per launch); " % int(os.environ.get("TH_BENCH_BATCH", "100000")) +
                   "valu_busy_upper = VALU wave-instructions x 4.2 cycles / (1024 SIMDs x 2.39 GHz x kernel seconds in this pass), salu_busy_upper likewise at 4.25 cycles (prices: profiles/r05_valu_rate.txt)", "kernels": out},
          open(os.path.join(O, "pmc_instr.json"), "w"), indent=1)
for k, v in out.items(): print("%-40s valu %.3e salu %.3e  %.3f s  valu busy <= %.2f  salu busy <= %.2f" % (k[:40], v["valu_wave_instr"], v["salu_wave_instr"], v["seconds_under_counters"], v["valu_busy_upper"] or 0, v["salu_busy_upper"] or 0))
PY
