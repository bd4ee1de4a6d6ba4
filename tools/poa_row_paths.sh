#!/bin/bash
# GPU box: instructions per row of each PATH of th_k_poa's row loop, MEASURED  ->  gpurun_out/poa_row_paths/poa_row_paths.json (copied to profiles/r06_poa_row_paths.json)
#   Build container first:  tools/build_variant.sh rdupm2 -DTH_POA_DUP_ROW=-2; for c in 0 1 2 3 4 5 6 7 9; do tools/build_variant.sh rdup$c -DTH_POA_DUP_ROW=$c; done
# Method: variant rdup<c> computes every row that takes path c TWICE (th_k_poa_body.h, TH_POA_DUP_ROW: the loop's scalars are restored in between and the row writes
#   the same values again; the records' digest must equal the product's), rdupm2 is the same code with no row repeated.  One rocprofv3 pass per variant with
#   --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES of `bench.py --quick --steps 1 --warmup 0 --digest`; (rdup<c> - rdupm2) / rows of path c (from
#   profiles/r06_poa_phases.json: the -DTH_POA_PROF=1 build's counts, same reads) = vector / scalar wave-instructions per row of that path, set-up, every trip and the
#   row's end included.  Check: sum over paths of rows x per-row against the DP phase's PMC totals of the same profile.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/poa_row_paths; mkdir -p $O; export TMPDIR=/tmp
E=$R/tests/cpu_emul
cd /tmp
for v in product rdupm2 rdup0 rdup1 rdup2 rdup3 rdup4 rdup5 rdup6 rdup7 rdup9; do
  L=$E/libth_$v.so; [ $v = product ] && L=$R/tidehunter_amd/libtidehunter_amd.so
  TH_BENCH_LIB=$L timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_$v -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick --digest > $O/pmc_$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); O = os.path.join(R, "gpurun_out/poa_row_paths")
res = {}
for v in ["product", "rdupm2"] + ["rdup%d" % c for c in (0, 1, 2, 3, 4, 5, 6, 7, 9)]:
    acc = collections.defaultdict(float); dur = 0.0
    for fn in glob.glob(os.path.join(O, "pmc_" + v, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "thpoa_n" in row["Kernel_Name"] and "th_k_poa<3072" in row["Kernel_Name"] and "short" in row["Kernel_Name"]: acc[row["Counter_Name"]] += float(row["Counter_Value"])
        os.remove(fn)
    for fn in glob.glob(os.path.join(O, "pmc_" + v, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "thpoa_n" in row["Kernel_Name"] and "th_k_poa<3072" in row["Kernel_Name"] and "short" in row["Kernel_Name"]: dur += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        os.remove(fn)
    dig = None
    for l in open(os.path.join(O, "pmc_%s.log" % v), errors="replace"):
        if l.startswith("{") and "records_digest" in l: dig = json.loads(l).get("records_digest")
    res[v] = {"valu": acc.get("SQ_INSTS_VALU"), "salu": acc.get("SQ_INSTS_SALU"), "wave_cycles": acc.get("SQ_WAVE_CYCLES"), "ms_under_counters": dur, "records_digest": dig}
    print(v, res[v])
json.dump(res, open(os.path.join(O, "raw.json"), "w"), indent=1)
PY
