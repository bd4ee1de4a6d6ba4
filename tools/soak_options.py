#!/usr/bin/env python3
"""GPU box: byte identity of the command line against the oracle's under OPTION sets, on more reads than the tests take and with seeds no test uses
(tests/test_gpu_parity.py runs these option sets on the reference's 16-read test.fq; tools/soak_scale.py runs the defaults at scale).
   usage: python3 tools/soak_options.py [reads per case = 4000]"""
import hashlib, os, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import tidehunter_amd as TH
import th_util as U
import oracle_lib as O
O.load()
EXE = os.environ.get("TH_SOAK_EXE") or os.path.join(R, "tidehunter_amd", "TideHunter-amd"); ORACLE = os.path.join(O.ODIR, "tho_oracle")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
lib = TH.load_library()
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
# two inputs: headline-shaped reads of mixed periods (seed 21) and adapter-bounded units of mixed periods on both strands (config 5 at spec, seed 22)
plain = os.path.join(tmp, "plain.fa"); adapt = os.path.join(tmp, "adapt.fa")
U.write_fasta(plain, [TH.synth_read(lib, 21, i, (150, 400, 1000, 2500)[i & 3], (30, 12, 10, 4)[i & 3], (0.05, 0.10, 0.13, 0.16)[(i >> 2) & 3]) for i in range(n)], prefix="so_")
reads5, ad_fns = U.config5_spec(tmp, n, seed=22)
U.write_fasta(adapt, reads5, prefix="sa_")
ad = ["-5", ad_fns[0], "-3", ad_fns[1]]
bad = 0
for fn, opts in [(plain, ["-f", "2"]), (plain, ["-u"]), (plain, ["-l"]), (plain, ["-l", "-u", "-f", "2"]), (plain, ["-c", "3", "-m", "100"]), (plain, ["-p", "100", "-P", "2K"]), (plain, ["-p", "300", "-e", "0.2"]),
                 (plain, ["-k", "6", "-w", "1", "-H"]), (plain, ["-k", "12", "-w", "4"]), (plain, ["-M", "1", "-X", "3", "-O", "3", "-E", "2"]), (plain, ["-r", "0.5", "-f", "4"]), (plain, ["-c", "5", "-f", "3"]),
                 (adapt, ad), (adapt, [*ad, "-F"]), (adapt, [*ad, "-F", "-s"]), (adapt, [*ad, "-F", "-f", "3"]), (adapt, [*ad, "-a", "0.6", "-F", "-f", "2"]), (adapt, [*ad, "-u", "-F"]), (adapt, [*ad, "-l", "-s", "-f", "2"])]:
    t0 = time.time(); gpu = subprocess.run([EXE, *opts, fn], capture_output=True, check=True).stdout; t1 = time.time()
    cpu = subprocess.run([ORACLE, "-t", str(U.effective_cpus()), *opts, fn], capture_output=True, check=True).stdout; t2 = time.time()
    same = gpu == cpu; bad += not same
    shown = " ".join(os.path.basename(o) if o.startswith(tmp) else o for o in opts)
    print("%-9s %5d reads  %-34s %s  %7d lines  md5 %s  gpu %.1f s  oracle %.1f s" % (os.path.basename(fn), n, shown, "IDENTICAL" if same else "DIFFERENT", gpu.count(b"\n"), hashlib.md5(gpu).hexdigest(), t1 - t0, t2 - t1), flush=True)
print("%d option sets, %d differ" % (19, bad))
sys.exit(1 if bad else 0)
