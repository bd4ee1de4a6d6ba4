#!/bin/bash
# helper for development runs on the GPU box: parity tests, then one bench line reduced to value + per-kernel ms
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 250 python bench.py --cpu-sample 0 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
