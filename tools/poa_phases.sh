#!/bin/bash
# GPU box: where th_k_poa's instructions and time go, per phase and per row category  ->  gpurun_out/poa_phases/poa_phases.json
#   (copied to profiles/r05_poa_phases.json).  Build container first:
#     tools/build_variant.sh base; tools/build_variant.sh prof1 -DTH_POA_PROF=1; for k in 0 1 2 3 9; do tools/build_variant.sh dup$k -DTH_POA_DUP_PHASE=$k; done
#   then tools/poa_phases_report.py > profiles/r05_poa_phases.json (build container: it reads the ISA)
# Method.  (1) PHASES: variant dup<k> runs phase k (0 row tables, 1 DP, 2 traceback, 3 consensus; all idempotent) twice inside the job loop, in a loop
#     whose trip count the compiler cannot see, so one copy of the code; per variant one rocprofv3 pass with --kernel-trace --pmc SQ_INSTS_VALU
#     SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES of `bench.py --quick --steps 1 --warmup 0` and one un-profiled bench line for the kernel time.
#     phase k = dup<k> - dup9 (the same code with no phase repeated); merge + job set-up + seed = base - the four.
# (2) ROW CATEGORIES of the DP: the prof1 variant (-DTH_POA_PROF=1: events only) counts rows per path of the row loop (one LDS add per row), further trips, traceback
#     events and the row-table phase's chunks and relaxation rounds.  (-DTH_POA_PROF=2 adds s_memtime cycles per path; that build faults on the GPU -- not used.)
# (3) STATIC: tools/isa_blocks.py counts of the straight-line row bodies are added by tools/poa_phases_report.py in the build container (it has the ISA).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/poa_phases; mkdir -p $O; export TMPDIR=/tmp
E=$R/tests/cpu_emul
cd $R
for v in base dup9 dup0 dup1 dup2 dup3; do
  for rep in 1 2; do
  TH_BENCH_LIB=$E/libth_$v.so timeout 600 python3 bench.py --steps 3 --warmup 1 --quick 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v', json.dumps({k: round(v,2) for k,v in j['roofline']['kernel_ms_per_step'].items()}))" >> $O/times.txt
  done
done
cd /tmp
for v in base dup9 dup0 dup1 dup2 dup3; do
  TH_BENCH_LIB=$E/libth_$v.so timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$v -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_$v.log 2>&1
done
cd $R
for v in prof1; do
  TH_BENCH_LIB=$E/libth_$v.so timeout 600 python3 bench.py --steps 1 --warmup 1 --quick 2> $O/$v.err > $O/$v.out
  grep "poa_prof" $O/$v.err > $O/$v.txt; tail -c 600 $O/$v.err | grep -i "fault\|error" 
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/poa_phases")
res = {}
for v in ("base", "dup9", "dup0", "dup1", "dup2", "dup3"):
    acc = collections.defaultdict(float); n = set(); dur = 0.0
    for fn in glob.glob(os.path.join(O, "pmc_" + v, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "thpoa_n" in row["Kernel_Name"] and "th_k_poa" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n.add(row["Dispatch_Id"])
        os.remove(fn)
    for fn in glob.glob(os.path.join(O, "pmc_" + v, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "thpoa_n" in row["Kernel_Name"] and "th_k_poa" in row["Kernel_Name"]:
                dur += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e9
        os.remove(fn)
    res[v] = dict(acc); res[v]["launches"] = len(n); res[v]["seconds_under_counters"] = dur
times = collections.defaultdict(list)
for l in open(os.path.join(O, "times.txt")):
    v, j = l.split(" ", 1); times[v].append(json.loads(j).get("th_k_poa"))
prof = {}
for v in ("prof1",):
    try:
        ls = [json.loads(l.split("[poa_prof] ", 1)[1]) for l in open(os.path.join(O, v + ".txt")) if "[poa_prof] {" in l]
        prof[v] = ls[-1] if ls else None
        lr = [json.loads(l.split("[poa_prof_rows] ", 1)[1]) for l in open(os.path.join(O, v + ".txt")) if "[poa_prof_rows] {" in l]
        prof[v + "_rows"] = lr[-1] if lr else None
    except OSError:
        prof[v] = None
json.dump({"pmc": res, "poa_ms": times, "prof_events_build": prof.get("prof1"), "prof_rows": prof.get("prof1_rows"), "prof_cycles_build": prof.get("prof")}, open(os.path.join(O, "poa_phases_raw.json"), "w"), indent=1)
print(json.dumps({"pmc": res, "poa_ms": times}, indent=1)); print(prof)
PY
