"""Kernel-logic parity on the CPU: the SAME kernel sources (tidehunter_amd/csrc/*.hip) compiled under the
test-only lock-step emulation (tests/cpu_emul/) against the oracle.  Small inputs only: the
emulation runs every HIP thread as a fiber.  The product library is never involved here; the GPU
parity proper is tests/test_gpu_parity.py (-m gpu)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import th_util as U
import tidehunter_amd as TH


@pytest.fixture(scope="module")
def emu():
    return U.emul_lib_path()


def _reads(lib, spec, seed=11):
    return [TH.synth_read(lib, seed, i, per, cop, err) for i, (per, cop, err) in enumerate(spec)]


def test_emul_test_50x4_golden(emu):
    th = TH.TideHunter(lib_path=emu)
    seq = U.read_fastx(os.path.join(U.DATA, "test_50x4.fa"))[0][1]
    (recs,) = th.process([seq])
    assert len(recs) == 1
    r = recs[0]     # README.md:222
    assert r["seq"] == b"CGATCGATCGGCATGCATGCATGCTAGTCGATGCATCGGGATCAGCTAGT"
    assert (r["start"] + 1, r["end"] + 1, r["cons_len"], r["full_length"]) == (51, 250, 50, 0)
    assert "%.1f %.1f" % (r["copy_num"], r["ave_match"]) == "4.0 100.0"
    assert [p + 2 for p in r["sub_pos"][:-1]] + [r["sub_pos"][-1] + 1] == [59, 109, 159, 208]
    th.close()


def test_emul_stages_and_records_small_synthetic(emu):
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12), (200, 3, 0.10), (90, 6, 0.0)])
    seqs += [b"ACGT" * 3, b"A" * 200, b"ACGTNNNNACGT" * 20, b""]          # ragged / degenerate inputs
    para = O.default_para()
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "emul") >= 5
    th.close()


def test_emul_unit_mode_and_filters(emu):
    spec = [(60, 8, 0.10), (120, 5, 0.08)]
    for kw in (dict(only_unit=1), dict(min_len=70), dict(only_longest=1), dict(min_copy=4), dict(out_fmt=3), dict(min_cov=3, out_fmt=4)):
        th = TH.TideHunter(lib_path=emu, **kw)
        seqs = _reads(th.lib, spec)
        para = O.default_para(**kw)
        U.assert_records_equal(seqs, th.process(seqs), para, str(kw))
        th.close()


def test_emul_bridging_alignment_sparse_anchors(emu):
    """large k makes anchors sparse, so the partition walk has to bridge gaps with the DP engine"""
    th = TH.TideHunter(lib_path=emu, k=12)
    seqs = _reads(th.lib, [(150, 5, 0.12), (150, 5, 0.15)], seed=5)
    para = O.default_para(k=12)
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    U.assert_records_equal(seqs, got, para, "k12")
    th.close()


def test_emul_adapters_full_length(emu):
    ad = (os.path.join(U.DATA, "5prime.fa"), os.path.join(U.DATA, "3prime.fa"))
    five = U.read_fastx(ad[0])[0][1]; three = U.read_fastx(ad[1])[0][1]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rng = np.random.default_rng(3)
    ins = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 70))
    unit = five + ins + three.translate(comp)[::-1]
    read = unit * 4
    th = TH.TideHunter(lib_path=emu, adapters=ad)
    para = O.default_para()
    O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
    got = th.process([read])
    assert got[0] and got[0][0]["full_length"] == 1 and got[0][0]["seq"] == ins
    U.assert_records_equal([read], got, para, "adapters")
    th.close()


def test_emul_hpc_kmers(emu):
    """-H: homopolymer-compressed k-mers (src/tandem_hit.c:46), incl. N resets and long runs"""
    th = TH.TideHunter(lib_path=emu, hpc=1)
    seqs = _reads(th.lib, [(80, 7, 0.08), (150, 4, 0.10)], seed=9)
    seqs += [(b"ACGGGGGTTTACGTNAACCCGTAGGA" * 3 + b"TTTTTTTTTT") * 5, b"AAAACCCCGGGGTTTT" * 30]
    para = O.default_para(hpc=1)
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    U.assert_records_equal(seqs, got, para, "hpc")
    th.close()


def test_emul_minimizer_seeding(emu):
    """-w > 1: the minimizer replay (src/tandem_hit.c:97-157), with N resets, ties in low-complexity sequence, and -H"""
    for kw in (dict(w=5), dict(w=3, k=6), dict(w=4, hpc=1), dict(w=10, k=5), dict(w=3, k=11), dict(k=10)):   # k > 9: 64-bit sort words
        th = TH.TideHunter(lib_path=emu, **kw)
        seqs = _reads(th.lib, [(80, 7, 0.08), (150, 4, 0.10)], seed=11)
        seqs += [(b"ACGGGGGTTTACGTNAACCCGTAGGA" * 3 + b"TTTTTTTTTT") * 5, b"ACACACACACAC" * 30, b"ACGT", b"ACGTACGTACGTNNACGTACGTACGTACGT"]
        para = O.default_para(**kw)
        got = th.process(seqs)
        U.stage_compare(th, seqs, para)
        U.assert_records_equal(seqs, got, para, "minimizer %r" % (kw,))
        th.close()


def test_emul_config5_adapters(emu, tmp_path):
    """adapter-bounded units on both strands through the host-side adapter search (edlib restatement)"""
    seqs, ad = U.config5_reads(tmp_path, 2, periods=(200, 240))
    for kw in (dict(), dict(only_full_length=1)):
        para = O.default_para(**kw)
        O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
        th = TH.TideHunter(lib_path=emu, adapters=ad, **kw)
        U.assert_records_equal(seqs, th.process(seqs), para, "config5 %r" % (kw,))
        th.close()


def test_emul_poa_scoring_options(emu):
    """-M -X -O -E: the packed int16 POA rows read their scores from a per-row table (match, mismatch <= 31); larger
    scores take the 32-bit rows"""
    for kw in (dict(match=3, mismatch=5), dict(gap_open1=6, gap_ext1=3, gap_open2=30, gap_ext2=1), dict(match=40, mismatch=50)):
        th = TH.TideHunter(lib_path=emu, **kw)
        seqs = _reads(th.lib, [(120, 6, 0.10), (260, 4, 0.12)], seed=17)
        U.assert_records_equal(seqs, th.process(seqs), O.default_para(**kw), "scoring %r" % (kw,))
        th.close()


def test_emul_chain_record_arena_grows(emu, monkeypatch):
    """a batch with more chains than the record arena holds reruns the extraction with a larger arena (TH_REC_CAP forces it)"""
    monkeypatch.setenv("TH_REC_CAP", "1")
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(90, 6, 0.08), (200, 4, 0.1)], seed=29)
    para = O.default_para()
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "arena growth") >= 2
    th.close()


def test_emul_job_tables_grow(emu, monkeypatch):
    """job / unit / par_pos tables that are too small for a batch (TH_JOB_CAP forces it) raise a flag in the scan, nothing is written,
    and the build reruns with tables sized from the totals; all three slow paths together as well"""
    monkeypatch.setenv("TH_JOB_CAP", "1")
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(90, 6, 0.08), (200, 4, 0.1), (60, 9, 0.1)], seed=33)
    para = O.default_para()
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "job table growth") >= 3
    th.close()
    monkeypatch.setenv("TH_REC_CAP", "1"); monkeypatch.setenv("TH_PAR_MUL", "1")
    th = TH.TideHunter(lib_path=emu)
    assert U.assert_records_equal(seqs, th.process(seqs), para, "all arenas too small") >= 3
    th.close()


def test_emul_partition_regions_grow(emu, monkeypatch):
    """a chain whose par_pos outgrows its region makes the partition rerun with larger regions (TH_PAR_MUL forces a first
    pass with four-entry regions)"""
    monkeypatch.setenv("TH_PAR_MUL", "1")
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(90, 6, 0.08), (200, 4, 0.1)], seed=31)
    para = O.default_para()
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    assert U.assert_records_equal(seqs, got, para, "partition regrow") >= 2
    th.close()


def test_emul_unpacked_chain_records(emu, monkeypatch):
    """batches holding a read of 65536+ bases use the unpacked chain-DP records; TH_CHAIN_UNPACKED forces that path on a
    small batch (the real trigger is covered on the GPU: test_gpu_batch_with_a_long_read)"""
    monkeypatch.setenv("TH_CHAIN_UNPACKED", "1")
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(90, 6, 0.08), (200, 4, 0.1)], seed=23)
    para = O.default_para()
    got = th.process(seqs)
    U.stage_compare(th, seqs, para)
    U.assert_records_equal(seqs, got, para, "unpacked chain records")
    th.close()


def test_refuses_out_of_range_window(emu):
    for kw in (dict(w=0), dict(w=257)):
        with pytest.raises(RuntimeError):
            TH.TideHunter(lib_path=emu, **kw)


def test_emul_banded_alignment_long_units(emu):
    """units long enough for th_k_align's certified banded pass (64 + 2W + |D| < 3/4 of the consensus length)"""
    th = TH.TideHunter(lib_path=emu)
    seqs = [TH.synth_read(th.lib, 5, 0, 500, 4, 0.10), TH.synth_read(th.lib, 5, 2, 520, 3, 0.16)]
    para = O.default_para()
    assert U.assert_records_equal(seqs, th.process(seqs), para, "band") >= 1
    th.close()


@pytest.mark.parametrize("wadd", ["24", "200", "330"])
def test_emul_antidiagonal_band_engine(emu, monkeypatch, wadd):
    """th_ksw_band_ad_wave (one anti-diagonal of the band per wave, packed int16 pairs): TH_ALN_MODE=2 recomputes every
    alignment with the full matrix on the device and fails the run on a difference; TH_ALN_WADD widens the requested band so
    that 1, 2 and 3 registers per lane are used; units of 330-500 bases reach the steady-state steps (band inside the
    matrix), one read carries Ns (the N-aware variant)"""
    monkeypatch.setenv("TH_ALN_MODE", "2")
    monkeypatch.setenv("TH_ALN_WADD", wadd)
    th = TH.TideHunter(lib_path=emu)
    seqs = [TH.synth_read(th.lib, 5, i, p, c, e) for i, (p, c, e) in enumerate([(400, 4, 0.10), (330, 5, 0.12), (500, 3, 0.05)])]
    s = bytearray(seqs[0])
    for x in (100, 101, 555, 900):
        s[x] = ord("N")
    seqs.append(bytes(s))
    assert U.assert_records_equal(seqs, th.process(seqs), O.default_para(), "anti-diagonal band, wadd " + wadd) >= 4
    th.close()


def test_emul_strip_engine_still_matches(emu, monkeypatch):
    """TH_ALN_MODE=0: the strip engine (th_ksw_global_band_wave), which remains the fallback for pairs the anti-diagonal
    engine cannot take"""
    monkeypatch.setenv("TH_ALN_MODE", "0")
    th = TH.TideHunter(lib_path=emu)
    seqs = [TH.synth_read(th.lib, 5, 0, 500, 4, 0.10), TH.synth_read(th.lib, 5, 2, 520, 3, 0.16)]
    assert U.assert_records_equal(seqs, th.process(seqs), O.default_para(), "strip") >= 1
    th.close()


def test_emul_poa_band_failure_is_rerun_without_a_band(emu, monkeypatch):
    """TH_POA_TEST_FAIL makes every job report "band lost the path" on the banded pass: the pipeline must rerun those jobs with
    full rows instead of failing the batch, and (the band never matters at these error rates) reach the same records"""
    monkeypatch.setenv("TH_POA_TEST_FAIL", "1")
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12)])
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "noband rerun") >= 3
    th.close()


def test_emul_poa_node_capacity_overflow_is_rerun(emu, monkeypatch):
    """round 4: a job's node arrays are carved for an estimate of its graph (1.5 x the longest unit plus 9 % per further unit), not for
    one node per base of every unit; TH_POA_TEST_SMALLCAP carves them just above the longest unit on the first attempt, so every graph
    outgrows them at its first merge: status 1, the rerun with twice the factor must reach the same records"""
    monkeypatch.setenv("TH_POA_TEST_SMALLCAP", "1")
    th = TH.TideHunter(lib_path=emu)
    seqs = _reads(th.lib, [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12), (200, 4, 0.15)])
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "node capacity rerun") >= 4
    th.close()


def test_emul_radix_sort_of_the_kmer_words(emu, monkeypatch):
    """round 4: th_k_hits sorts its 32-bit words (key << 14 | position, written in position order) with a stable LSD radix sort of the
    key field -- two passes of 8-bit digits, per-wave counters, lanes of a step ranked with ballots -- instead of the bitonic network.
    Hits must equal the oracle's (src/tandem_hit.c:171-225) and the bitonic path's (TH_SORT_BITONIC=1) on reads that fill 2, 8 and 32
    steps per wave, with N bases (sentinel words inside the array), homopolymer stretches (many equal keys) and -k 9 (a third pass of
    two bits) and -k 6 (a second pass of four)"""
    import numpy as np
    lib = TH.load_library(emu)
    rng = np.random.default_rng(3)
    seqs = [TH.synth_read(lib, 21, 0, 700, 14, 0.08), TH.synth_read(lib, 21, 1, 300, 7, 0.10), TH.synth_read(lib, 21, 2, 97, 9, 0.05)]
    s = bytearray(TH.synth_read(lib, 21, 3, 450, 9, 0.10))
    for x in rng.integers(0, len(s), 60):
        s[int(x)] = ord("N")
    s[1000:1300] = b"A" * 300
    s[2000:2200] = b"AC" * 100
    seqs.append(bytes(s))
    for kw in ({}, {"k": 9}, {"k": 6}, {"hpc": 1}):
        got = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("TH_SORT_BITONIC", mode)
            th = TH.TideHunter(lib_path=emu, only_unit=1, **kw)
            th.process(seqs)
            U.stage_compare(th, seqs, O.default_para(only_unit=1, **kw))
            T = sum(len(x) for x in seqs)
            got[mode] = (th.debug_stage(1, 4 * len(seqs), np.int32).copy(), th.debug_stage(2, 4 * T, np.int32).copy(), th.debug_stage(3, 4 * T, np.int32).copy())
            th.close()
        assert all(np.array_equal(a, b) for a, b in zip(got["0"], got["1"])), kw


def test_emul_file_driver_pipeline(emu, tmp_path):
    """th_run_file_multi on the emulation library: reader -> device pipelines (upload / run / collect threads each; one or two per
    device, batches taken from the shared ticket counter, ramped batch sizes) -> writer; several batches, output order = input order, bytes equal to the oracle's command line"""
    import ctypes as C
    import subprocess
    th = TH.TideHunter(lib_path=emu, out_fmt=2)
    seqs = _reads(th.lib, [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12), (80, 6, 0.05), (50, 9, 0.1), (70, 7, 0.1), (64, 6, 0.08)])
    fa = tmp_path / "in.fa"
    U.write_fasta(fa, seqs)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]; libc.fclose.argtypes = [C.c_void_p]
    th.lib.th_run_file_multi.argtypes = [C.c_char_p, C.POINTER(TH.Para), C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int]
    want = subprocess.run([os.path.join(O.ODIR, "tho_oracle"), "-t", "2", "-f", "2", str(fa)], check=True, capture_output=True).stdout
    import gzip
    gz, bgz = tmp_path / "in.fa.gz", tmp_path / "in.fa.bgz"
    with gzip.open(gz, "wb") as f:
        f.write(fa.read_bytes())
    U.write_bgzf(bgz, fa.read_bytes(), block=300)
    for devs, batch, env, src in (([0], 2, {}, fa), ([0, 0], 3, {}, fa), ([0], 100, {"TH_RAMP0": "1"}, fa), ([0], 2, {"TH_WORKERS_PER_DEV": "2"}, fa),
                                  ([0], 3, {}, gz), ([0], 3, {}, bgz)):
        for k, v in env.items():
            os.environ[k] = v
        try:
            out = tmp_path / "out.txt"
            fp = libc.fopen(str(out).encode(), b"w")
            arr = (C.c_int * len(devs))(*devs)
            n = th.lib.th_run_file_multi(str(src).encode(), th.para, fp, arr, len(devs), batch)
            libc.fclose(fp)
        finally:
            for k in env:
                del os.environ[k]
        assert n == len(seqs), (devs, batch, env)
        assert out.read_bytes() == want, (devs, batch, env)
    th.close()


def test_emul_pipelined_batches_and_sync_count(emu):
    """th_process_batches(): five batches (one empty, one repeated) through the upload / kernels / record-arithmetic pipeline give the
    records of five separate th_process_batch() calls; a run takes exactly three host synchronisations (after the job tables, after
    the POA, at the end) -- the host builds no table in the middle of the path"""
    import ctypes as C
    th = TH.TideHunter(lib_path=emu)
    a = _reads(th.lib, [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12)])
    b = _reads(th.lib, [(80, 6, 0.05), (50, 9, 0.1)], seed=12)
    c = _reads(th.lib, [(70, 7, 0.1), (64, 6, 0.08), (90, 5, 0.1), (45, 10, 0.1)], seed=13)
    batches = [a, b, [], c, a]
    want = [th.process(x) if x else [] for x in batches]
    assert th.syncs() == 3
    nb = len(batches)
    pp, ip = C.POINTER(C.c_char_p), C.POINTER(C.c_int)
    keep = [((C.c_char_p * max(len(x), 1))(*x), (C.c_int * max(len(x), 1))(*[len(s) for s in x])) for x in batches]
    outs = [th.lib.th_alloc_tandem_seq(max(len(x), 1)) for x in batches]
    rc = th.lib.th_process_batches(th.h, nb, (pp * nb)(*[C.cast(k[0], pp) for k in keep]), (ip * nb)(*[C.cast(k[1], ip) for k in keep]),
                                   (C.c_int * nb)(*[len(x) for x in batches]), (C.POINTER(TH.TandemSeq) * nb)(*outs))
    assert rc == 0
    for x, o, w in zip(batches, outs, want):
        th._ts, th._n, saved = o, len(x), th._ts
        got = th.records()
        th._ts = saved
        assert got == w
        th.lib.th_free_tandem_seq(o, max(len(x), 1))
    th.close()


def test_ticket_segment_starts_from_zero_after_a_killed_run(emu):
    """a creator that finds a segment of the same name (left by a killed run, counters advanced) must not inherit its values"""
    lib = TH.load_library(emu)
    name = b"/th_test_ticket_%d" % os.getpid()
    t = lib.th_ticket_open(name, 1)
    assert [lib.th_ticket_next(t, 0) for _ in range(3)] == [0, 1, 2]
    lib.th_ticket_close(t, 0)                      # (no unlink: what a killed run leaves behind)
    t = lib.th_ticket_open(name, 1)
    peer = lib.th_ticket_open(name, 0)
    assert lib.th_ticket_next(t, 0) == 0 and lib.th_ticket_next(peer, 0) == 1 and lib.th_ticket_next(t, 5) == 0
    lib.th_ticket_close(peer, 0); lib.th_ticket_close(t, 1)


def test_emul_parallel_minimizer_rules(emu, monkeypatch):
    """-w > 1: the wave-parallel selection (per-position rules) against the oracle's restatement of src/tandem_hit.c:97-157 and
    against the sequential replay kept in the kernel (TH_MINIMIZER_SEQ=1); TH_MINIMIZER_STRICT fails the run if any read had to
    fall back to the replay.  Inputs with N runs, runs shorter than a window, low-complexity stretches, w up to 256."""
    rng = np.random.default_rng(7)
    rnd = lambda n, alpha=b"ACGT": bytes(alpha[i] for i in rng.integers(0, len(alpha), n))
    lib = TH.load_library()
    seqs = [TH.synth_read(lib, 5, i, 80 + 7 * i, 5, 0.08) for i in range(4)]
    seqs += [rnd(300, b"AC"), rnd(400, b"ACGTN"), b"ACGT" * 60, b"A" * 150 + rnd(100), rnd(40), rnd(12), b"ACGTNACGTNNNACGTACGTACGTTTGACAAC" * 9,
             rnd(150) + b"N" + rnd(9) + b"N" + rnd(200), rnd(3), b""]
    monkeypatch.setenv("TH_MINIMIZER_STRICT", "1")
    for kw in (dict(w=5), dict(w=2, k=8), dict(w=40, k=6), dict(w=256, k=3), dict(w=7, k=16)):
        para = O.default_para(**kw)
        for seq_only in (False, True):
            if seq_only:
                monkeypatch.setenv("TH_MINIMIZER_SEQ", "1")
            th = TH.TideHunter(lib_path=emu, **kw)
            monkeypatch.delenv("TH_MINIMIZER_SEQ", raising=False)
            th.process(seqs)
            U.stage_compare(th, seqs, para)
            th.close()


def test_emul_chain_dp_lds_ring_wraps(emu, monkeypatch):
    """reads with more hits than the LDS ring of the chain DP holds (512): the ring wraps, scores are patched in place; with and
    without the ring the DP equals the oracle's (stages only: -u skips POA, which is what makes 10 kb reads slow in the emulation)"""
    seqs = [TH.synth_read(TH.load_library(), 9, 0, 300, 8, 0.08), TH.synth_read(TH.load_library(), 9, 1, 150, 14, 0.1), TH.synth_read(TH.load_library(), 9, 2, 60, 6, 0.1)]
    for ring in ("512", "nolists", "640", "768", "1024", "0"):
        monkeypatch.setenv("TH_CHAIN_RING", "512" if ring == "nolists" else ring)
        monkeypatch.setenv("TH_CHAIN_CLS", "0" if ring == "nolists" else "1")
        th = TH.TideHunter(lib_path=emu, only_unit=1)
        th.process(seqs)
        assert int(th.debug_stage(1, 4 * len(seqs), np.int32)[0]) > 1000
        U.stage_compare(th, seqs, O.default_para(only_unit=1))
        th.close()


def test_emul_device_adapter_search_block_boundaries(emu, tmp_path):
    """th_k_adapter (Myers bit-vector, 64-symbol blocks): adapter lengths on both sides of the block boundaries, lower-case and N
    in the adapters, a looser -a, and an adapter of more than 256 symbols (kept on the host); records equal the oracle's edlib restatement"""
    rng = np.random.default_rng(21)
    rnd = lambda n: bytes(b"ACGT"[i] for i in rng.integers(0, 4, n))
    comp = bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN")
    rc = lambda s: s.translate(comp)[::-1]
    for l5, l3, ada in ((63, 5, 0.8), (64, 65, 0.8), (129, 30, 0.7), (84, 25, 0.9), (300, 40, 0.8)):
        five, three = rnd(l5), rnd(l3)
        five = five[:10].lower() + b"N" + five[11:] if l5 > 20 else five
        f5, f3 = tmp_path / ("5_%d.fa" % l5), tmp_path / ("3_%d.fa" % l3)
        f5.write_bytes(b">five\n" + five + b"\n"); f3.write_bytes(b">three\n" + three + b"\n")
        seqs = []
        for i in range(4):
            unit = five.upper().replace(b"N", b"A") + rnd(40 + 10 * i) + rc(three)
            s = bytearray()
            for b in unit * 4:
                x = rng.random()
                if x < 0.03:
                    s.append(b"ACGT"[int(rng.integers(0, 4))])
                elif x < 0.05:
                    continue
                else:
                    s.append(b)
            seqs.append(rc(bytes(s)) if i & 1 else bytes(s))
        para = O.default_para(ada_match_rat=ada)
        O.load().tho_para_load_adapters(para, str(f5).encode(), str(f3).encode())
        th = TH.TideHunter(lib_path=emu, adapters=(str(f5), str(f3)), ada_match_rat=ada)
        got = th.process(seqs)
        assert U.assert_records_equal(seqs, got, para, "adapters %d/%d" % (l5, l3)) >= 3
        if l5 <= 256:
            assert any(r["full_length"] > 0 for rr in got for r in rr), (l5, l3)
        th.close()


def test_emul_batch_runs_as_two_concurrent_halves(emu, monkeypatch):
    """a large batch is cut into two halves by bases that run on two pipelines of the handle at once (TH_SUBBATCH_MIN lowers the
    threshold for the test): same records as the oracle, merged in read order; adapters and -s included"""
    monkeypatch.setenv("TH_SUBBATCH_MIN", "3")
    # (three parts as well)
    monkeypatch.setenv("TH_SUBBATCH", "3")
    th = TH.TideHunter(lib_path=emu)
    seqs3 = _reads(TH.load_library(), [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12), (200, 3, 0.10), (90, 6, 0.0)])
    got3 = th.process(seqs3)
    assert th.launches() == 3
    U.assert_records_equal(seqs3, got3, O.default_para(), "three parts")
    th.close()
    monkeypatch.setenv("TH_SUBBATCH", "2")
    ad = (os.path.join(U.DATA, "5prime.fa"), os.path.join(U.DATA, "3prime.fa"))
    seqs = _reads(TH.load_library(), [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12), (200, 3, 0.10), (90, 6, 0.0), (70, 7, 0.1)]) + [b"ACGT" * 3, b""]
    for kw, adapters in ((dict(), None), (dict(out_fmt=4, min_cov=2), None), (dict(only_full_length=1, single_copy=1), ad)):
        th = TH.TideHunter(lib_path=emu, adapters=adapters, **kw)
        got = th.process(seqs)
        assert th.launches() == 2
        para = O.default_para(**kw)
        if adapters:
            O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
        U.assert_records_equal(seqs, got, para, "two halves %r" % (kw,))
        assert th.process(seqs[:2]) == got[:2] and th.launches() == 1
        th.close()


def test_emul_poa_relative_score_variant(monkeypatch):
    """thpoa_l, the long-unit POA configuration (int16 scores relative to a running base, ring rows of up to 384 columns in three trips):
    forced on short units (TH_POA_FORCE_LONG) in a build whose rebasing threshold is 40 instead of 8000, so that the ring is rebased every
    few rows, upwards and downwards -- records equal the oracle's, including far rows (full cells in absolute scores) and bubbles"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(U.ROOT, "tests", "cpu_emul"), "libth_emul_rel.so"], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(U.ROOT, "tests", "cpu_emul", "libth_emul_rel.so")
    monkeypatch.setenv("TH_POA_FORCE_LONG", "1")
    th = TH.TideHunter(lib_path=lib)
    seqs = _reads(th.lib, [(60, 8, 0.10), (120, 5, 0.08), (35, 12, 0.12), (200, 4, 0.15), (90, 6, 0.0), (150, 7, 0.2), (300, 4, 0.1)])
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "relative-score POA") >= 6
    th.close()
    # unusual scoring: larger steps, more rebases
    th = TH.TideHunter(lib_path=lib, match=5, mismatch=9, gap_open1=8, gap_ext1=4, gap_open2=30, gap_ext2=2)
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(match=5, mismatch=9, gap_open1=8, gap_ext1=4, gap_open2=30, gap_ext2=2), "relative-score POA, other scoring") >= 6
    th.close()
    # the same build carries the opt-in 8-bit direction words (TH_POA_D8=1: rows with one predecessor): the ordinary int16 variant with them
    monkeypatch.delenv("TH_POA_FORCE_LONG")
    th = TH.TideHunter(lib_path=lib)
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "8-bit direction words") >= 6
    th.close()


def test_emul_poa_long_variant_three_trip_rows(emu):
    """a read of 2100-base units (9 copies): its POA runs on thpoa_l with bands of more than 256 columns, i.e. three 128-column trips per
    row, some starting at an odd column with the ring's full width (the case a ring of 3 x 128 columns got wrong: a band of RW columns
    from an odd start needs RW + 2 slots); records equal the oracle's"""
    seq = U.read_fastx(os.path.join(U.DATA, "long_unit_2100x9.fa"))[0][1]
    th = TH.TideHunter(lib_path=emu)
    got = th.process([seq])
    assert U.assert_records_equal([seq], got, O.default_para(), "2100 x 9") == 1
    th.close()


def test_emul_flank_extension_engines(emu, monkeypatch):
    """flank extensions (ksw2_left_ext / ksw2_right_ext, src/gen_cons.c:218-221; full matrix, no z-drop) on their three exact engines: a
    period-1000 read at 10 % error (two registers per lane certified by th_adx_bound, the bound that also counts the matches an
    off-diagonal path gives up), a period-1400 read at 15 % (the first band fails, its banded maximum names the one that holds) and a
    period-2300 read at 16 % (no register band: a four-register probe sizes a band of the strip engine); TH_ALN_MODE=2 repeats the
    first two on the full matrix on the device and flags a difference; records equal the oracle's"""
    monkeypatch.setenv("TH_ALN_MODE", "2")
    th = TH.TideHunter(lib_path=emu)
    seqs = [TH.synth_read(th.lib, 21, 0, 1000, 4, 0.10), TH.synth_read(th.lib, 22, 1, 1400, 3, 0.15)]
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "flank extensions, full matrix alongside") >= 1
    th.close()
    monkeypatch.delenv("TH_ALN_MODE")
    th = TH.TideHunter(lib_path=emu)         # (the banded strip engine against the oracle only: the emulation takes minutes for a second full matrix)
    seqs = [TH.synth_read(th.lib, 23, 2, 2300, 4, 0.16)]
    got = th.process(seqs)
    assert U.assert_records_equal(seqs, got, O.default_para(), "flank extension on a band of the strip engine") == 1
    th.close()


def test_emul_single_copy_reads_searched_on_the_device():
    """-F -s (src/gen_cons.c:128-171): reads that carry the adapters once -- sense, antisense, with errors, adapters twice, lower case --
    get their full-length record from th_k_adapter_reads (whole-read Myers searches over the uploaded bytes, up to three per query as
    collect_ed_res does); records equal the oracle's"""
    rng = np.random.default_rng(5)
    rnd = lambda n: bytes(b"ACGT"[i] for i in rng.integers(0, 4, n))
    comp = bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN")
    rc = lambda s: s.translate(comp)[::-1]
    ad = (os.path.join(U.DATA, "5prime.fa"), os.path.join(U.DATA, "3prime.fa"))
    five, three = U.read_fastx(ad[0])[0][1], U.read_fastx(ad[1])[0][1]
    def mut(s, n):
        s = bytearray(s)
        for _ in range(n):
            s[int(rng.integers(0, len(s)))] = b"ACGT"[int(rng.integers(0, 4))]
        return bytes(s)
    unit = five + rnd(300) + rc(three)
    seqs = [rnd(40) + unit + rnd(25), rc(rnd(10) + unit + rnd(70)), rnd(5) + mut(five, 3) + rnd(250) + rc(mut(three, 1)) + rnd(5),
            rnd(30) + unit + rnd(90) + unit + rnd(12), (rnd(20) + unit + rnd(20)).lower(), rnd(400), five + rnd(20) + rc(three), b"ACGT" * 2]
    kw = dict(only_full_length=1, single_copy=1)
    for extra in (dict(), dict(out_fmt=4), dict(ada_match_rat=0.7)):
        th = TH.TideHunter(lib_path=U.emul_lib_path(), adapters=ad, **kw, **extra)
        got = th.process(seqs)
        para = O.default_para(**kw, **extra)
        O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
        n = U.assert_records_equal(seqs, got, para, "single copy %r" % (extra,))
        assert n >= 4, n
        th.close()


def test_emul_chain_dp_period_class_lists_everywhere():
    """round 6: th_k_chain_dp's period-class lists (th_k_chain.hip, CLS) -- a head of a listed period bin scans the hits of its own chunk, then its class's list of
    finished hits instead of the hit array.  The product lists bins with 16 heads, long windows and lists that filter; this build (`libth_emul_cls.so`: two heads, any
    window, any share) lists EVERYTHING, so the listed scan -- own-chunk trip, list trips, first / last trips of a list, the max_h rule (src/tandem_chain.c:349) on hit
    indices in dense windows of error-free and low-error short periods, more bins than lists -- runs on every shape; scores and predecessors of every hit, chains and
    par_pos against the oracle's stage dumps (src/tandem_chain.c:325-356)"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(U.ROOT, "tests", "cpu_emul"), "libth_emul_cls.so"], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(U.ROOT, "tests", "cpu_emul", "libth_emul_cls.so")
    L = TH.load_library()
    rng = np.random.default_rng(17)
    seqs = [TH.synth_read(L, 31, i, p, c, e) for i, (p, c, e) in enumerate([(1000, 6, 0.10), (64, 60, 0.02), (64, 60, 0.0), (40, 90, 0.01), (200, 20, 0.10), (300, 12, 0.0),
                                                                             (131, 30, 0.05), (2500, 3, 0.12), (97, 40, 0.15), (500, 9, 0.08)])]
    # a read of several different repeats in a row: many period bins with heads (more than the 24 lists a read may keep)
    seqs.append(b"".join(TH.synth_read(L, 32, i, p, 5, 0.06) for i, p in enumerate(range(70, 1400, 55))))
    seqs.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, 4000)))
    th = TH.TideHunter(lib_path=lib, only_unit=1)
    th.process(seqs)
    U.stage_compare(th, seqs, O.default_para(only_unit=1))
    th.close()
