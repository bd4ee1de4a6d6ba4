# development helper (GPU box): alignment-kernel time for several settings of its environment knobs, one process, one batch
#   python tools/aln_variants.py 32768 1000 10 0.1 "TH_ALN_WTOL=0" "TH_ALN_WTOL=16" "TH_ALN_MODE=0" "LIB=path/to/other/build.so" ...
import os, sys, time
sys.path.insert(0, '.')
import tidehunter_amd as TH
lib = TH.load_library()
B, PERIOD, COPIES, ERR = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
seqs = [TH.synth_read(lib, 1, i % 100000, PERIOD, COPIES, ERR) for i in range(B)]
for spec in sys.argv[5:]:
    kv = dict(x.split("=") for x in spec.split(",") if x)
    for k, v in kv.items(): os.environ[k] = v
    th = TH.TideHunter(device=0, lib_path=kv.pop("LIB")) if "LIB" in kv else TH.TideHunter(device=0)
    for rep in range(2):
        th.upload(seqs); t0 = time.time(); th.run(); th.collect_native(); t1 = time.time()
    print("%-40s step %.1f ms  %.0f reads/s " % (spec, 1000 * (t1 - t0), B / (t1 - t0)), {k: round(v, 1) for k, v in th.timing().items()}, flush=True)
    th.close()
    for k in kv: del os.environ[k]
