"""Randomized byte-identity sweeps of the command line against the CPU oracle's, inside `pytest -m gpu` (round 1 ran them as
shell scripts, tools/parity_sweep*.sh): shapes, error rates, N content, lower case, partial last copies, flanks and option
sets that the fixed tests do not cover -- 63 input x option combinations without adapters, 20 with adapters (-5/-3, -F, -s,
-a, -r), FASTA and gzipped FASTQ.  Everything POA touches is "GPU == oracle" (abPOA is absent from the reference tree)."""
import gzip
import os
import random
import subprocess

import pytest

import oracle_lib as O
import th_util as U

pytestmark = pytest.mark.gpu

EXE = os.path.join(U.ROOT, "tidehunter_amd", "TideHunter-amd")
ORACLE = os.path.join(O.ODIR, "tho_oracle")
SEED_SHIFT = int(os.environ.get("TH_SWEEP_SEED", "0"))      # soak runs: other inputs of the same shapes (tools/soak.sh)


def _noisy(seq, err, pn, rng):
    out = []
    for b in seq:
        x = rng.random()
        if x < err / 3:
            out.append(rng.choice("ACGT".replace(b, "")))
        elif x < 2 * err / 3:
            out.append(b); out.append(rng.choice("ACGT"))
        elif x < err:
            pass
        else:
            out.append(b)
        if pn and out and rng.random() < pn:
            out[-1] = "N"
    return "".join(out)


def _both(opts, fn):
    O.load()
    gpu = subprocess.run([EXE, *opts, str(fn)], capture_output=True)
    cpu = subprocess.run([ORACLE, "-t", str(U.effective_cpus()), *opts, str(fn)], capture_output=True)
    assert gpu.returncode == 0 and cpu.returncode == 0, (opts, gpu.stderr[-300:], cpu.stderr[-300:])
    return gpu.stdout, cpu.stdout


SHAPES = [("p50", 50, 30, 0.05, 0.0, 300), ("p137_e0", 137, 12, 0.0, 0.0, 200), ("p733_e20", 733, 6, 0.20, 0.0, 200),
          ("p2500", 2500, 4, 0.12, 0.0, 100), ("p3500_wide", 3500, 4, 0.10, 0.0, 80), ("p9000", 9000, 3, 0.10, 0.0, 24),
          ("p400_N", 400, 8, 0.08, 0.01, 200), ("p1000_c2", 1000, 3, 0.10, 0.0, 200), ("mixed", 0, 0, 0.10, 0.002, 300)]
OPTS = [[], ["-f", "2"], ["-f", "4", "-c", "2"], ["-k", "11", "-p", "20"], ["-w", "4", "-H"], ["-e", "0.1", "-l"],
        ["-M", "3", "-X", "5", "-O", "5,30", "-E", "3,1"]]


@pytest.fixture(scope="module")
def sweep_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("sweep")
    rng = random.Random(2024 + SEED_SHIFT)
    files = {}
    for name, period, copies, err, pn, n in SHAPES:
        fn = d / ("sw_%s.fa" % name)
        with open(fn, "w") as f:
            for i in range(n):
                p = period or rng.choice([31, 64, 200, 999, 1500, 2100])
                c = copies or rng.randint(2, 12)
                unit = "".join(rng.choice("ACGT") for _ in range(p))
                flank5 = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 300)))
                flank3 = "".join(rng.choice("acgt") for _ in range(rng.randint(0, 300)))
                f.write(">%s_%d\n%s\n" % (name, i, flank5 + _noisy(unit * c, err, pn, rng) + unit[:rng.randint(0, p - 1)] + flank3))
        files[name] = fn
    return files


@pytest.mark.parametrize("shape", [s[0] for s in SHAPES])
def test_gpu_randomized_sweep(sweep_files, shape, monkeypatch):
    monkeypatch.setenv("TH_MINIMIZER_STRICT", "1")
    for opts in OPTS:
        gpu, cpu = _both(opts, sweep_files[shape])
        assert gpu == cpu, "%s %r: output differs from the oracle's" % (shape, opts)


def test_gpu_randomized_sweep_adapters(tmp_path):
    rng = random.Random(77 + SEED_SHIFT)
    rc = lambda s: s.translate(str.maketrans("ACGT", "TGCA"))[::-1]
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
    five, three = rnd(84), rnd(25)
    f5, f3 = tmp_path / "sw5.fa", tmp_path / "sw3.fa"
    f5.write_text(">5p\n%s\n" % five); f3.write_text(">3p\n%s\n" % three)
    recs = []
    for i in range(400):
        p = rng.choice([200, 350, 600, 1200, 2000])
        unit = five + rnd(p - 109) + rc(three)
        c = max(2, 4000 // p) if i % 7 else 1          # some single-copy reads for -s
        s = rnd(rng.randint(0, 60)) + _noisy(unit * c, rng.choice([0.03, 0.1, 0.15]), 0.0, rng) + rnd(rng.randint(0, 60))
        recs.append(("ad_%d" % i, rc(s) if i & 1 else s))
    fa, fq = tmp_path / "sw_ad.fa", tmp_path / "sw_ad.fq.gz"
    with open(fa, "w") as f:
        for n, s in recs:
            f.write(">%s\n%s\n" % (n, s))
    with gzip.open(fq, "wt", compresslevel=1) as f:
        for n, s in recs:
            f.write("@%s some comment\n%s\n+\n%s\n" % (n, s, "".join(rng.choice("!#5AIJ") for _ in s)))
    A = ["-5", str(f5), "-3", str(f3)]
    for fn in (fa, fq):
        for opts in ([], A, A + ["-F"], A + ["-F", "-s"], A + ["-f", "2", "-a", "0.7"], A + ["-f", "4", "-r", "0.5"], ["-r", "3", "-f", "3"],
                     ["-u"], ["-u", "-f", "2", "-m", "300"], A + ["-F", "-l", "-c", "3"]):
            gpu, cpu = _both(opts, fn)
            assert gpu == cpu, "%s %r: output differs from the oracle's" % (fn.name, opts[4:] if opts[:1] == ["-5"] else opts)
