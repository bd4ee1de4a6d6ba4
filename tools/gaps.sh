#!/bin/bash
# development helper: idle time between consecutive kernels of one bench step (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/gaps; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $O/log 2>&1
python3 - <<'PY'
import csv, os, glob
fn = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/gaps/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 0]
half = rows[len(rows) // 2:]           # the timed step (second half of the dispatches)
prev = None
for r in half:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e6 if prev else 0.0
    print("%-40s dur %9.3f ms   gap before %8.3f ms" % (r["Kernel_Name"][:40], (e - s) / 1e6, gap))
    prev = e
os.remove(fn)
PY
