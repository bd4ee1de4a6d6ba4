# development helper: per-step kernel times over several different batches (variance check)
import sys, time
sys.path.insert(0, '.')
import tidehunter_amd as TH
th = TH.TideHunter(device=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
PERIOD = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
COPIES = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ERR = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
batches = [[TH.synth_read(th.lib, 1, (b * B + i) % 100000, PERIOD, COPIES, ERR) for i in range(B)] for b in range(3)]
for rep in range(4):
    seqs = batches[rep % 3]
    th.upload(seqs); t0 = time.time(); th.run(); th.collect_native(); t1 = time.time()
    print("batch %d  step %.1f ms  %.0f reads/s " % (rep % 3, 1000 * (t1 - t0), B / (t1 - t0)), {k: round(v, 1) for k, v in th.timing().items()})
