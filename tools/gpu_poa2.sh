#!/bin/bash
# development helper (GPU box): bench + POA-heavy parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-poa2}; mkdir -p $O; cd $R
timeout 900 python bench.py --cpu-sample 0 --e2e 0 --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
python3 -c "import json; j=json.load(open('$O/bench.json')); print(round(j['value']), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_launch'].items()})"
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
