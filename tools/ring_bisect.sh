cd ${GRAFT_REPO_ROOT:-/root/repo}
cp tidehunter_amd/libtidehunter_amd.so /tmp/prod.so
trap 'cp /tmp/prod.so tidehunter_amd/libtidehunter_amd.so' EXIT      # (an interrupted run must not leave a variant in the product's place: ADVICE r5)
for L in ${LIBS:-libth_base.so libth_skipu.so libth_ch2.so libth_ch3.so}; do
  cp tests/cpu_emul/$L tidehunter_amd/libtidehunter_amd.so
  for ring in 512 0 768 1024; do
  r=$(TH_ONLY_RING=$ring timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 100 -k "ring_variants" 2>&1 | grep -E "AssertionError|passed|failed" | head -3 | tr '\n' ' ')
  echo "$L ring $ring: $r"
  done
done
cp /tmp/prod.so tidehunter_amd/libtidehunter_amd.so
