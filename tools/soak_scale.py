#!/usr/bin/env python3
"""GPU box: byte identity of the command line against the oracle's on MORE reads than the tests take, drawn with other seeds
(test_gpu_scale.py: 4096 / 4096 / 512 reads of seeds 1 / 2 / 3).   usage: python3 tools/soak_scale.py [scale] [case,case...]"""
import hashlib, os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import tidehunter_amd as TH
import th_util as U
import oracle_lib as O
O.load()
EXE = os.path.join(R, "tidehunter_amd", "TideHunter-amd"); ORACLE = os.path.join(O.ODIR, "tho_oracle")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None       # optional: names of the cases to run
lib = TH.load_library()
for name, n, period, copies, err, seed, opts in [("C3", 20000, 1000, 10, 0.10, 11, []), ("C2", 20000, 200, 20, 0.10, 12, ["-f", "2"]),
                                                 ("C4", 2000, 5000, 4, 0.15, 13, []), ("p2000", 4000, 2000, 5, 0.12, 14, ["-f", "2"]),
                                                 ("long", 160, 2000, 50, 0.10, 15, [])]:       # (100 kb reads: the HBM sorts, the unpacked chain DP, thpoa_l with 50 units per job)
    if only and name not in only:
        continue
    n = max(64, int(n * scale))
    fn = "/dev/shm/soak_%s.fa" % name
    U.write_fasta(fn, [TH.synth_read(lib, seed, i, period, copies, err) for i in range(n)], prefix="soak_%s_" % name)
    t0 = time.time(); gpu = subprocess.run([EXE, *opts, fn], capture_output=True, check=True).stdout; t1 = time.time()
    cpu = subprocess.run([ORACLE, "-t", str(U.effective_cpus()), *opts, fn], capture_output=True, check=True).stdout; t2 = time.time()
    print("%-6s %6d reads (seed %d) %s: %s  records %d  md5 %s  gpu %.1f s  oracle %.1f s" % (
        name, n, seed, " ".join(opts) or "-", "IDENTICAL" if gpu == cpu else "DIFFERENT", gpu.count(b">"), hashlib.md5(gpu).hexdigest(), t1 - t0, t2 - t1), flush=True)
    # round 6: the same reads through the product library and its two test twins -- built WITHOUT -structurizecfg-skip-uniform-regions, and with the chain DP's
    # period-class lists for every bin -- via th_process_batch + th_format_output: the three must print the command line's bytes
    reads = [s for _, s in U.read_fastx(fn)]
    names = [b"soak_%s_%d" % (name.encode(), i) for i in range(n)]
    for tag, path in (("product", None), ("flag-off twin", os.path.join(R, "tests", "cpu_emul", "libth_nostruct.so")), ("lists-everywhere twin", os.path.join(R, "tests", "cpu_emul", "libth_clsall.so"))):
        if path is not None and not os.path.exists(path):
            print("       %s: not built" % tag); continue
        t = TH.TideHunter(device=0, lib_path=path, out_fmt=2 if "-f" in opts else 1)
        t.process(reads); text = t.format_output(names); t.close()
        print("       %-22s md5 %s %s" % (tag, hashlib.md5(text).hexdigest(), "= the command line" if text == gpu else "DIFFERS FROM THE COMMAND LINE"), flush=True)
    os.unlink(fn)
