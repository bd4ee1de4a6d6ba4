#!/bin/bash
# development helper (GPU box): GPU tests + config benches
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-check4}; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
CONFIGS="${CONFIGS:-c3 c5}" E2E=0 bash tools/gpu_configs.sh $1
