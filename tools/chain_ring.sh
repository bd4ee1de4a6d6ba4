#!/bin/bash
# development helper (GPU box): chain-DP time per LDS ring size (TH_CHAIN_RING) at the headline shape, and the kernel's HBM fetch traffic for each
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ring; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for ring in ${1:-512 768 1024}; do
  export TH_CHAIN_RING=$ring
  python3 $R/bench.py --quick --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring $ring', round(j['value']), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_step'].items()})"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$ring -o run -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/f$ring.log 2>&1
  python3 - $O/f$ring <<'PY'
import csv, glob, sys, os
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for row in csv.DictReader(open(fn)):
        if row["Counter_Name"] == "FETCH_SIZE" and "chain_dp" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]] = acc.get(row["Kernel_Name"].split("(")[0], 0.0) + float(row["Counter_Value"])
    for k, v in acc.items(): print("   ", k, "FETCH_SIZE %.1f GB raw, %.1f GB corrected (x2)" % (v * 1024 / 1e9, 2 * v * 1024 / 1e9))
    os.remove(fn)
PY
done
