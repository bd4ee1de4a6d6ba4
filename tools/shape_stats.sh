#!/bin/bash
# GPU box: rocprofv3 kernel statistics of one bench step of another shape (c2 / c4 / c5): which launches the time goes to
#   usage: tools/shape_stats.sh "c4 c5"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/shape_stats; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for c in ${1:-c4 c5}; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -o run -- python3 $R/bench.py --config $c --steps 2 --warmup 1 --quick > $O/$c.log 2>&1
  f=$(find $O/$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %4s total %9.1f ms avg %9.2f ms  %5s %%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
  cp "$f" $O/${c}_kernel_stats.csv
  python3 - "$(find $O/$c -name "*kernel_trace.csv" | head -1)" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "th_k_align" in r["Kernel_Name"] or "th_k_poa" in r["Kernel_Name"]]
print("launch order of the last step (ms):", "  ".join("%s %.1f" % (r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in sel[-(len(sel) // 3):]))
PY
  find $O/$c -name "*kernel_trace.csv" -delete
done
