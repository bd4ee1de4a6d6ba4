# GPU box: th_k_chain_dp with the period-class lists off and on (TH_CHAIN_CLS=0 / 1): VALU / SALU wave-instructions, wave-cycles, vector-memory and LDS instructions per launch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc3; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  TH_CHAIN_CLS=$v timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/pmc_$v -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_$v.log 2>&1
  python3 - $O/pmc_$v cls$v <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float)
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)): acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]]+=float(r["Counter_Value"])
for fn in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)): dur[r["Kernel_Name"].split("(")[0][-60:]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,v in acc.items():
    if "chain_dp" in k: print(sys.argv[2], k, "ms %.1f"%dur[k], {c: round(x/1e9,2) for c,x in v.items()})
PY
done
