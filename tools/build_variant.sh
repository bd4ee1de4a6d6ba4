#!/bin/bash
# development helper: build the working tree's kernels as tests/cpu_emul/libth_<name>.so (for TH_BENCH_LIB / tools/gpu_variants2.sh)
# and print the POA kernel's register / spill / instruction statistics from its ISA.   usage: tools/build_variant.sh name [-Dflags]
set -e
R=$(cd $(dirname $0)/.. && pwd); N=$1; shift
W=/tmp/th_variant_$N; mkdir -p $W; cd $R/tidehunter_amd/csrc
F="$(make -s --no-print-directory print-hipflags) -I. -I../../include $*"      # the product's flags (csrc/Makefile) + the variant's
/opt/rocm/bin/hipcc $F -c th_pipeline.hip -o $W/pipe.o -save-temps=obj 2>$W/hipcc.err || { grep -m5 -A3 "error" $W/hipcc.err; exit 1; }
gcc -O2 -g -std=gnu99 -fPIC -Wall -Wno-misleading-indentation -I. -I../../include -c th_host.c -o $W/host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tests/cpu_emul/libth_$N.so $W/pipe.o $W/host.o -lz -lm -lpthread
S=$(ls $W/*gfx950*.s | head -1)
python3 - "$S" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r'^(_ZN7thpoa_n8th_k_poaILi3072EsLb1E[^:\n]*):\s.*?^\s*\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel', txt, re.S | re.M)
body = m.group(0)
ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith((';', '.'))]
v = sum(i.startswith('v_') for i in ins); s = sum(i.startswith('s_') for i in ins)
lane = sum(i.startswith(('v_readlane', 'v_writelane')) for i in ins)
spill = len(re.findall(r'v_(?:readlane_b32 s\d+, v7\d, \d+|writelane_b32 v7\d, s\d+, \d+)', body))
g = lambda k: re.search(r'\.amdhsa_%s (\d+)' % k, body)
print("th_k_poa<3072,short>: static VALU %d SALU %d lane-ops %d (spill-like %d) next_free_vgpr %s sgpr %s scratch %s" % (
    v, s, lane, spill, g('next_free_vgpr').group(1), g('next_free_sgpr').group(1), g('private_segment_fixed_size').group(1)))
PY
python3 $R/tools/isa_check_saddr_hazard.py "$S" | tail -2
