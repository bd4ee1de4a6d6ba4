# development helper: per-step kernel times over several different batches (variance check)
import sys, time
sys.path.insert(0, '.')
import tidehunter_amd as TH
th = TH.TideHunter(device=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
batches = [[TH.synth_read(th.lib, 1, (b * B + i) % 100000, 1000, 10, 0.1) for i in range(B)] for b in range(3)]
for rep in range(7):
    seqs = batches[rep % 3]
    th.upload(seqs); t0 = time.time(); th.run(); th.collect_native(); t1 = time.time()
    print("batch %d  step %.1f ms " % (rep % 3, 1000 * (t1 - t0)), {k: round(v, 1) for k, v in th.timing().items()})
