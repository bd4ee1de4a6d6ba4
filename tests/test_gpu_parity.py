"""GPU parity proper: the HIP path (tidehunter_amd/libtidehunter_amd.so, through the C-ABI) against the
CPU oracle on the same inputs.  Bit-exact: consensus strings, every integer field, and the doubles
(which both sides compute with the same expressions from the same integers).  For everything POA
touches the claim is "GPU == oracle/tho_poa.c", not "== reference" (abPOA is absent)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import th_util as U
import tidehunter_amd as TH

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def th():
    t = TH.TideHunter(device=0)
    yield t
    t.close()


def _synth(lib, n, period, copies, err, seed):
    return [TH.synth_read(lib, seed, i, period, copies, err) for i in range(n)]


def test_gpu_readme_line_222(th):
    seq = U.read_fastx(os.path.join(U.DATA, "test_50x4.fa"))[0][1]
    (recs,) = th.process([seq])
    r = recs[0]
    assert r["seq"] == b"CGATCGATCGGCATGCATGCATGCTAGTCGATGCATCGGGATCAGCTAGT"
    assert (r["start"] + 1, r["end"] + 1, r["cons_len"], r["full_length"]) == (51, 250, 50, 0)
    assert "%.1f %.1f" % (r["copy_num"], r["ave_match"]) == "4.0 100.0"
    assert [p + 2 for p in r["sub_pos"][:-1]] + [r["sub_pos"][-1] + 1] == [59, 109, 159, 208]


def test_gpu_reference_test_data(th):
    para = O.default_para()
    seqs = [U.read_fastx(os.path.join(U.DATA, f))[0][1] for f in ("test_50x4.fa", "test_1000x10.fa", "full_length.fa")]
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "test_data") >= 3


def test_gpu_test_fq_real_reads(th):
    para = O.default_para()
    seqs = [s for _, s in U.read_fastx(os.path.join(U.DATA, "test.fq"))]
    assert len(seqs) == 100
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "test.fq") == 57     # 57 groups


@pytest.mark.parametrize("period,copies,err,n,seed", [
    (200, 20, 0.10, 256, 2),     # config 2 shape: short period, chain-dominated
    (1000, 10, 0.10, 256, 1),    # config 3 shape (headline)
    (5000, 4, 0.15, 32, 3),      # config 4 shape: long period, wide POA band, 5000^2 alignments
    (50, 30, 0.05, 16, 9),
])
def test_gpu_synthetic_configs(th, period, copies, err, n, seed):
    para = O.default_para()
    seqs = _synth(th.lib, n, period, copies, err, seed)
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "p%d" % period) >= n // 2


def test_gpu_ragged_and_degenerate_inputs(th):
    para = O.default_para()
    rng = np.random.default_rng(0)
    seqs = [b"", b"ACG", b"ACGTACGT", b"A" * 5000, b"AC" * 3000, b"ACGTNNNNACGT" * 50,
            bytes(b"ACGT"[i] for i in rng.integers(0, 4, 3000)),          # no repeat at all
            b"acgtnacgt-acgt" * 40]
    seqs += _synth(th.lib, 3, 300, 6, 0.1, 4)
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    U.assert_records_equal(seqs, got, para, "ragged")


def test_gpu_options(th):
    seqs = _synth(th.lib, 6, 150, 8, 0.1, 21)
    for kw in (dict(only_unit=1), dict(min_len=160), dict(only_longest=1), dict(min_copy=5), dict(out_fmt=3),
               dict(min_cov=4, out_fmt=4), dict(k=12), dict(max_div=0.05), dict(min_p=200), dict(max_p=100), dict(hpc=1), dict(hpc=1, k=6), dict(w=5), dict(w=10, k=5), dict(w=4, hpc=1),
               dict(match=3, mismatch=5), dict(gap_open1=6, gap_ext1=3, gap_open2=30, gap_ext2=1), dict(match=40, mismatch=50), dict(mismatch=1, gap_ext1=1)):
        t = TH.TideHunter(device=0, **kw)
        U.assert_records_equal(seqs, t.process(seqs), O.default_para(**kw), str(kw))
        t.close()


def test_gpu_batch_with_a_long_read(th):
    """a read of 65536+ bases switches the whole batch to the unpacked chain-DP records, the 64-bit sort words and the
    HBM sort scratch"""
    rng = np.random.default_rng(5)
    long_read = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 66000))
    seqs = [long_read] + _synth(th.lib, 4, 90, 6, 0.08, 23) + _synth(th.lib, 2, 1000, 5, 0.1, 24)
    para = O.default_para()
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    U.assert_records_equal(seqs, got, para, "long read in batch")


def test_gpu_adapters_full_length(th):
    ad = (os.path.join(U.DATA, "5prime.fa"), os.path.join(U.DATA, "3prime.fa"))
    para = O.default_para()
    O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
    seq = U.read_fastx(os.path.join(U.DATA, "full_length.fa"))[0][1]
    t = TH.TideHunter(device=0, adapters=ad)
    got = t.process([seq])
    assert got[0][0]["full_length"] == 1 and got[0][0]["start"] + 1 == 31 and got[0][0]["end"] + 1 == 5215
    U.assert_records_equal([seq], got, para, "full_length")
    t.close()


def test_gpu_config5_adapters(tmp_path):
    """config 5 shape: adapter-bounded units of mixed periods, both strands, with and without -F (full length only)"""
    seqs, ad = U.config5_reads(tmp_path, 24)
    for kw in (dict(), dict(only_full_length=1), dict(only_full_length=1, single_copy=1)):
        para = O.default_para(**kw)
        O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
        t = TH.TideHunter(device=0, adapters=ad, **kw)
        got = t.process(seqs)
        n = U.assert_records_equal(seqs, got, para, "config5 %r" % (kw,))
        assert n >= 12
        if not kw:
            assert sum(1 for rr in got for r in rr if r["full_length"] > 0) >= 8
        t.close()


def test_gpu_batch_order_and_idempotence(th):
    """results do not depend on what else is in the batch or on running it twice"""
    seqs = _synth(th.lib, 12, 250, 6, 0.1, 33)
    a = th.process(seqs)
    b = th.process(seqs[::-1])[::-1]
    c = th.process(seqs)
    assert a == b == c


def test_gpu_headline_shape_properties(th):
    """size-independent properties at the headline read shape: every read yields one consensus of
    roughly one period, spanning nearly the whole read, with ~10 copies"""
    seqs = _synth(th.lib, 64, 1000, 10, 0.10, 1)
    got = th.process(seqs)
    one = [r for r in got if len(r) >= 1]
    assert len(one) >= 60
    for recs in one:
        r = max(recs, key=lambda x: x["end"] - x["start"])
        assert 900 <= r["cons_len"] <= 1100 and r["end"] - r["start"] > 8000 and 8.5 <= r["copy_num"] <= 11.5
        assert r["ave_match"] > 80.0


def _cli(*args):
    import subprocess
    exe = os.path.join(U.ROOT, "tidehunter_amd", "TideHunter-amd")
    return subprocess.run([exe, *args], check=True, capture_output=True, text=True).stdout


def _oracle_cli(*args):
    import subprocess
    O.load()
    return subprocess.run([os.path.join(O.ODIR, "tho_oracle"), *args], check=True, capture_output=True, text=True).stdout


def test_gpu_cli_output_is_byte_identical():
    """the command line (reader -> batch seam -> writer) against the oracle's fixtures and its CLI, byte for byte"""
    fq = os.path.join(U.DATA, "test.fq")
    golden = os.path.join(U.ROOT, "tests", "golden")
    assert _cli("-f", "2", fq) == open(os.path.join(golden, "oracle_test_fq_tab.txt")).read()
    assert _cli("-u", "-f", "2", fq) == open(os.path.join(golden, "oracle_test_fq_unit.txt")).read()
    ad = ["-5", os.path.join(U.DATA, "5prime.fa"), "-3", os.path.join(U.DATA, "3prime.fa")]
    fl = os.path.join(U.DATA, "full_length.fa")
    assert _cli("-f", "3", *ad, fl) == open(os.path.join(golden, "oracle_full_length_fastq.txt")).read()
    for extra in ([], ["-f", "4"], ["-l"], ["-c", "3", "-m", "100"], ["-F", *ad], ["-s", "-F", *ad], ["-p", "100", "-P", "2K"], ["-u"]):
        assert _cli(*extra, fq) == _oracle_cli("-t", "8", *extra, fq), extra
    # README.md:222
    assert _cli("-f", "2", os.path.join(U.DATA, "test_50x4.fa")) == \
        "test_50x4\trep0\t4.0\t300\t51\t250\t50\t100.0\t0\t59,109,159,208\tCGATCGATCGGCATGCATGCATGCTAGTCGATGCATCGGGATCAGCTAGT\n"


def test_gpu_long_reads_take_the_hbm_sort_paths(th):
    """reads with > 16384 k-mers / > 8192 hits: bitonic sorts in HBM scratch, block-wide extraction kernel"""
    para = O.default_para()
    seqs = [TH.synth_read(th.lib, 13, 0, 2000, 20, 0.08), TH.synth_read(th.lib, 13, 1, 150, 200, 0.05), TH.synth_read(th.lib, 13, 2, 700, 6, 0.1)]
    assert len(seqs[0]) > 35000
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    U.assert_records_equal(seqs, got, para, "long")


def test_gpu_multi_batch_file_run(tmp_path):
    """th_run_file with a small batch size: several device batches, output order = input order"""
    fq = os.path.join(U.DATA, "test.fq")
    assert _cli("-b", "7", "-f", "2", fq) == open(os.path.join(U.ROOT, "tests", "golden", "oracle_test_fq_tab.txt")).read()


def test_gpu_variant_boundaries(th):
    """read lengths on both sides of the per-batch kernel-variant thresholds (32-bit sort words below 16384 / 32768 bases,
    packed chain-DP records below 65536): a tandem repeat embedded in a random read of exactly that length"""
    rng = np.random.default_rng(11)
    para = O.default_para()
    for L in (16383, 16384, 32767, 32768, 65535, 65536):
        core = _synth(th.lib, 1, 300, 6, 0.08, L)[0]
        pad = bytes(b"ACGT"[i] for i in rng.integers(0, 4, L - len(core)))
        seqs = [pad[:L // 3] + core + pad[L // 3:], _synth(th.lib, 1, 120, 5, 0.1, L + 1)[0]]
        assert len(seqs[0]) == L
        got = th.process(seqs)
        U.stage_compare(th, seqs, para)
        assert U.assert_records_equal(seqs, got, para, "L=%d" % L) >= 2


def test_gpu_multi_device_file_run():
    """th_run_file_multi: batches go round a device list (here two workers on the one GPU of the box), the writer restores
    input order; same bytes as the single-device run and as the oracle"""
    fq = os.path.join(U.DATA, "test.fq")
    want = open(os.path.join(U.ROOT, "tests", "golden", "oracle_test_fq_tab.txt")).read()
    assert _cli("-g", "0,0", "-b", "7", "-f", "2", fq) == want
    assert _cli("-g", "0,0,0", "-b", "64", "-f", "2", fq) == want


def test_gpu_large_batch_is_deterministic(th):
    """a few thousand reads, three runs: identical records every time (persistent kernels pull work from
    shared queues and chain records arrive in arbitrary order; nothing downstream may depend on that)"""
    seqs = _synth(th.lib, 3072, 1000, 10, 0.10, 1)
    runs = [th.process(seqs) for _ in range(3)]
    assert runs[0] == runs[1] == runs[2]
    para = O.default_para()
    step = len(seqs) // 24
    sample = list(range(0, len(seqs), step))
    U.assert_records_equal([seqs[i] for i in sample], [runs[0][i] for i in sample], para, "sampled")


@pytest.mark.parametrize("env", [dict(TH_ALN_MODE="2"), dict(TH_ALN_MODE="2", TH_ALN_WADD="200"), dict(TH_ALN_MODE="2", TH_ALN_WADD="600"),
                                 dict(TH_ALN_MODE="2", TH_ALN_WTOL="0"), dict(TH_ALN_MODE="0"), dict(TH_ALN_MODE="2", TH_ALN_SLB="0"), dict(TH_ALN_SLB="0"),
                                 # (small batches take the longest-first instantiation of th_k_align, the headline's 358 items per wave the plain one -- both at this size)
                                 dict(TH_ALN_LPT="0"), dict(TH_ALN_LPT="1", TH_ALN_MODE="2")])
def test_gpu_alignment_engines(monkeypatch, env):
    """unit-vs-consensus alignments: the anti-diagonal band engine with 1..6 registers per lane (TH_ALN_WADD widens the
    requested band), verified on the device against the full matrix (TH_ALN_MODE=2 fails the run on any difference), and the
    strip engine alone (TH_ALN_MODE=0); the flank extensions likewise (register bands certified by th_adx_bound, the retry with the band
    the banded maximum names, the banded strip engine behind a four-register probe for the 2500- and 5000-base units at 15 % error);
    records compared with the oracle as everywhere else"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t = TH.TideHunter(device=0)
    para = O.default_para()
    seqs = _synth(t.lib, 12, 1000, 10, 0.10, 1) + _synth(t.lib, 6, 200, 20, 0.10, 2) + _synth(t.lib, 3, 2500, 4, 0.15, 3) + _synth(t.lib, 4, 700, 6, 0.2, 6) + _synth(t.lib, 2, 5000, 4, 0.15, 4)
    s = bytearray(seqs[0])
    for x in range(1500, 1510):
        s[x] = ord("N")
    seqs.append(bytes(s))
    got = t.process(seqs)
    assert U.assert_records_equal(seqs, got, para, "alignment engines %r" % (env,)) >= 20
    t.close()


@pytest.mark.parametrize("env", [dict(), dict(TH_ALN_MODE="2"), dict(TH_ALN_SLB="0")])
def test_gpu_long_units_start_from_the_poa_bound(monkeypatch, env):
    """round 5: th_k_poa leaves, per unit, the ksw2 score of the unit's own path through the graph against the consensus columns (a valid global alignment, so a
    lower bound of ksw2_global's optimum, src/gen_cons.c:213) and th_k_align runs long units on the band that bound certifies instead of probing for a bound.
    Units of 2500 ... 5000 bases at 10 ... 16 % on the relative-int16 and the 32-bit POA variants; TH_ALN_MODE=2 recomputes every result on the full matrix on
    the device, TH_ALN_SLB=0 is round 4's probing flow; a bound that is none fails the run (TH_E_ALN_BOUND).  Records equal the oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t = TH.TideHunter(device=0)
    seqs = _synth(t.lib, 10, 5000, 4, 0.15, 3) + _synth(t.lib, 6, 3000, 5, 0.12, 7) + _synth(t.lib, 4, 2500, 6, 0.16, 8) + _synth(t.lib, 4, 4000, 5, 0.10, 9) + _synth(t.lib, 2, 7000, 4, 0.12, 10)
    got = t.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "POA bounds %r" % (env,)) >= 20
    assert t.stats()["jobs_without_consensus"] == 0
    t.close()


def test_gpu_two_concurrent_halves_match_one_pipeline(monkeypatch):
    """TH_SUBBATCH=2: a batch of 16384 reads or more runs as two concurrent halves (two streams, two sets of buffers) and is merged by
    read index: same records as one pipeline (the default), and as the oracle on a sample; mixed read lengths so that the halves are cut
    by bases, not by count"""
    lib = TH.load_library()
    seqs = [TH.synth_read(lib, 41, i, (150, 300, 90)[i % 3], (6, 5, 10)[i % 3], 0.08) for i in range(18000)]
    monkeypatch.setenv("TH_SUBBATCH", "2")
    t2 = TH.TideHunter(device=0)
    a = t2.process(seqs)
    assert t2.launches() == 2
    t2.close()
    monkeypatch.setenv("TH_SUBBATCH", "1")
    t1 = TH.TideHunter(device=0)
    b = t1.process(seqs)
    assert t1.launches() == 1
    t1.close()
    assert a == b
    sample = list(range(0, len(seqs), 150))
    assert U.assert_records_equal([seqs[i] for i in sample], [a[i] for i in sample], O.default_para(), "two halves") >= 100


@pytest.mark.gpu
def test_gpu_chain_dp_ring_variants_agree(monkeypatch):
    """th_k_chain_dp with its LDS ring of 512 records (the default: next trip's records requested by an LDS read the compiler does not
    track) with and without the period-class lists of round 6, without a ring (every trip from HBM) and with the 640 / 768 / 1024-entry rings: identical hits, scores,
    predecessors, chains and par_pos on reads whose windows are longer than the ring, all against the oracle's stage dumps"""
    lib = TH.load_library()
    seqs = [TH.synth_read(lib, 9, i, (1000, 300, 2000)[i % 3], (10, 30, 5)[i % 3], 0.10) for i in range(96)]
    para = O.default_para(only_unit=1)
    for ring in (os.environ["TH_ONLY_RING"],) if os.environ.get("TH_ONLY_RING") else ("512", "nolists", "0", "640", "768", "1024"):      # (TH_ONLY_RING: tools/ring_bisect.sh)
        monkeypatch.setenv("TH_CHAIN_RING", "512" if ring == "nolists" else ring)
        monkeypatch.setenv("TH_CHAIN_CLS", "0" if ring == "nolists" else "1")      # (round 6: the default ring runs with the period-class lists; "nolists": every head scans the hit array)
        t = TH.TideHunter(device=0, only_unit=1)
        t.process(seqs)
        U.stage_compare(t, seqs, para)
        t.close()


def test_gpu_chain_dp_period_class_lists_everywhere():
    """round 6: the chain DP's period-class lists (th_k_chain.hip, CLS).  The product lists period bins with 16 heads, windows of 512 hits and lists that hold at most half
    of the read's hits -- so on the device its listed scan only ever sees the long windows of multi-period heads.  `tests/cpu_emul/libth_clsall.so` (csrc/Makefile `clsall`,
    built by build(); built here if the snapshot lacks it) is the same library with a list for EVERY bin that has two heads: the listed scan -- own-chunk trip from the
    LDS ring, list trips, the max_h rule (src/tandem_chain.c:349) on hit indices in dense windows, lists dropped at capacity, more bins than lists -- runs on every shape.
    Hits, DP scores and predecessors of every hit, chains and par_pos against the oracle's stage dumps (src/tandem_chain.c:325-356), and with the product library as well."""
    import subprocess
    ca = os.path.join(U.ROOT, "tests", "cpu_emul", "libth_clsall.so")
    if not os.path.exists(ca):
        subprocess.run(["make", "-s", "-C", os.path.join(U.ROOT, "tidehunter_amd", "csrc"), "clsall"], check=True, timeout=900)
    L = TH.load_library()
    rng = np.random.default_rng(23)
    seqs = _synth(L, 96, 1000, 10, 0.10, 41) + _synth(L, 64, 200, 20, 0.10, 42) + _synth(L, 12, 5000, 4, 0.15, 43) + _synth(L, 24, 2000, 5, 0.12, 44)
    seqs += [TH.synth_read(L, 45, i, p, c, e) for i, (p, c, e) in enumerate([(64, 60, 0.02), (64, 60, 0.0), (40, 90, 0.01), (300, 12, 0.0), (131, 30, 0.05), (97, 40, 0.15), (500, 9, 0.08), (31, 100, 0.03)] * 3)]
    seqs += [b"".join(TH.synth_read(L, 46 + r, i, p, 5, 0.06) for i, p in enumerate(range(70, 1400, 55))) for r in range(4)]        # many period bins with heads in one read
    seqs += [s for _, s in U.read_fastx(os.path.join(U.DATA, "test.fq"))]
    seqs.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, 30000)))
    para = O.default_para(only_unit=1)
    for path in (ca, None):
        t = TH.TideHunter(device=0, lib_path=path, only_unit=1)
        t.process(seqs)
        U.stage_compare(t, seqs, para)
        t.close()


def test_gpu_long_tandem_reads_take_the_long_read_paths():
    """bench.py's `c_long` point (2000 reads of 100 kb, period 2000 x 50: outside the contract) has no oracle beside it at that size; here six such reads -- 70 ... 100 kb,
    tandem repeats of 35 ... 50 units, so that the seeding's 64-bit sort words and HBM sort, the UNPACKED chain-DP records (reads of 65 536 bases and more: no period-class
    lists either), the wave extraction's large instantiation and POA jobs of 50 units all see real repeat structure, not the random bases of
    test_gpu_batch_with_a_long_read -- against the oracle: hits, DP cells, chains, par_pos and every record field."""
    L = TH.load_library()
    seqs = [TH.synth_read(L, 71, i, p, c, 0.10) for i, (p, c) in enumerate([(2000, 50), (2000, 35), (1500, 60), (2500, 30), (1000, 80), (3000, 25)])]
    assert all(len(s) >= 65536 for s in seqs)
    para = O.default_para()
    t = TH.TideHunter(device=0)
    got = t.process(seqs)
    U.stage_compare(t, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "long tandem reads") >= 6
    assert t.stats()["jobs_without_consensus"] == 0
    t.close()


def test_gpu_poa_scratch_factor_follows_the_batch_before():
    """Reads of 50 copies: the graph of 50 units outgrows the first attempt's scratch estimate (rows and band cells of a typical job x 1.5), the jobs are given back and rerun
    with twice, then four times the factor (th_poa_retries, th_pipeline.hip).  When most jobs of a class needed that, the handle's NEXT batch starts the class there
    (poa_fexp0): no job is given back again, and the records -- of both batches -- are the oracle's (the factor sizes arrays, nothing else).  TH_POA_ADAPT=0 keeps 1.5."""
    L = TH.load_library()
    seqs = [TH.synth_read(L, 72, i, 2000, 50, 0.10) for i in range(20)]
    para = O.default_para()
    t = TH.TideHunter(device=0)
    first = t.process(seqs)
    r1 = t.stats()["poa_jobs_rerun"]
    second = t.process(seqs)
    r2 = t.stats()["poa_jobs_rerun"]
    assert r1 >= 10, "the first batch no longer needs reruns: pick a shape that does (%d)" % r1
    assert r2 == r1, "the second batch gave %d jobs back" % (r2 - r1)
    assert first == second
    assert U.assert_records_equal(seqs, second, para, "50-copy reads, second batch") >= 20
    assert t.stats()["jobs_without_consensus"] == 0
    t.close()
    os.environ["TH_POA_ADAPT"] = "0"
    try:
        t = TH.TideHunter(device=0)
        assert t.process(seqs) == first and t.process(seqs) == first
        assert t.stats()["poa_jobs_rerun"] == 2 * r1
        t.close()
    finally:
        del os.environ["TH_POA_ADAPT"]
