#!/bin/bash
# GPU box: the randomized byte-identity sweeps (tests/test_gpu_sweep.py) on inputs drawn with other seeds, the last pass with every banded
# alignment verified against the full matrix on the device (TH_ALN_MODE=2).   usage: tools/soak.sh "1 2 3"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for s in ${1:-1 2 3}; do
  echo "== seed shift $s"; TH_SWEEP_SEED=$s timeout 1500 python3 -m pytest tests/test_gpu_sweep.py -q -m gpu -x 2>&1 | tail -3
done
echo "== seed shift 11, TH_ALN_MODE=2"; TH_SWEEP_SEED=11 TH_ALN_MODE=2 timeout 1500 python3 -m pytest tests/test_gpu_sweep.py -q -m gpu -x 2>&1 | tail -3
