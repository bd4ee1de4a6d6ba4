#!/bin/bash
# development helper: one bench line per value of an environment knob ($1 = name, rest = values)
name=$1; shift
for v in "$@"; do
  echo -n "$name=$v: "
  env $name=$v timeout 250 python bench.py --cpu-sample 0 --steps 2 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
done
