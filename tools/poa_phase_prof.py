import sys, time
sys.path.insert(0, '.')
import tidehunter_amd as TH
lib = TH.load_library()
B = 8192
seqs = [TH.synth_read(lib, 1, i, 1000, 10, 0.1) for i in range(B)]
th = TH.TideHunter(device=0, lib_path='tests/cpu_emul/libth_prof.so')
for rep in range(2):
    th.upload(seqs); th.run(); th.collect_native()
    print({k: round(v, 1) for k, v in th.timing().items()}, flush=True)
th.close()
