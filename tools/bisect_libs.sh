#!/bin/bash
# runs the reference-data parity check against each variant library by swapping it into place (GPU box scratch copy only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp tidehunter_amd/libtidehunter_amd.so /tmp/prod.so
trap 'cp /tmp/prod.so tidehunter_amd/libtidehunter_amd.so' EXIT      # (an interrupted run must not leave a variant in the product's place: ADVICE r5)
for L in "$@"; do
  cp tests/cpu_emul/$L tidehunter_amd/libtidehunter_amd.so
  r=$(timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 100 -k "readme or reference_test_data or real_reads" 2>&1 | tail -1)
  echo "$L: $r"
done
cp /tmp/prod.so tidehunter_amd/libtidehunter_amd.so
