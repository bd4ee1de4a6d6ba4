"""Known-answer tests of the HOST decision logic whose expectations come from READING THE REFERENCE, not from the oracle.

Why: the product's record arithmetic (tidehunter_amd/csrc/th_host.c) and the oracle's (oracle/tho_cons.c,
tho_io.c, tho_main.c) were written by the same hand from the same reference text, so "product == oracle" cannot see a transcription
error that both share.  Here every input is built so that the answer is known BY CONSTRUCTION, and the expectation is typed in the test
next to the reference line it follows:

  * a read is an ERROR-FREE tandem repeat of a random unit of length P.  The first repeated k-mer ends at base P + k - 1 with period P,
    so the first boundary is base k - 1 (src/tandem_chain.c:251-255, src/partition.c:171-276 on exact anchors) and every further one is
    P later: sub_pos = [k - 1 + i P].  All units are identical, so ANY consensus caller returns unit 0 -- the read rotated by k bases --
    whatever libabpoa does (POA parity is unpinned; these tests do not depend on it).  iden_n = len for every unit, so aveMatch = 100;
    the flank extensions match every base, so copyNum = read length / P exactly and the record spans the whole read
    (src/gen_cons.c:208-223).
  * adapters are exact copies (or copies with ONE substitution) placed at chosen offsets of that consensus, so every edit distance of
    src/gen_cons.c:226-291 is known: 0 or 1, and nothing else in a random unit comes within the threshold.

Each case runs through (i) the oracle, (ii) the product's host code over the CPU emulation of the kernels (tests/cpu_emul: th_host.c and
th_main.c are the product's files), and (iii) `-m gpu`: the product library / command line on the device.  The three must give the
hand-typed answer; they are never compared with each other here."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import th_util as U
import tidehunter_amd as TH

ROOT = U.ROOT
K = 8                                   # default k (src/tidehunter.h: DEF_KMER_LEN): the consensus is the unit rotated by K bases
_rng = np.random.default_rng(20261004)


def rnd(n, rng=None):
    return bytes(b"ACGT"[i] for i in (rng or _rng).integers(0, 4, n))


def seeded(seed):
    """a generator of a test's own: its draws do not depend on which tests ran before it in this process (pytest-xdist spreads the suite over workers)"""
    return np.random.default_rng(seed)


def is_rotation(a, b):
    return len(a) == len(b) and a in b + b


_comp = bytes.maketrans(b"ACGT", b"TGCA")


def rc(s):
    return s.translate(_comp)[::-1]


def sub1(s, i):
    """s with ONE substitution at i (edit distance exactly 1 against s)"""
    return s[:i] + bytes([b"ACGT"[(b"ACGT".index(s[i]) + 1) & 3]]) + s[i + 1:]


def read_of(cons, copies=8, tail=0):
    """the error-free read whose consensus (before adapter trimming) is `cons`: the unit is `cons` rotated back by K bases"""
    unit = cons[-K:] + cons[:-K]
    return unit * copies + unit[:tail]


# adapters of the test's own choosing (24 and 20 bases: thresholds (int)(24 * 0.2f) = 4 and (int)(20 * 0.2f) = 3, src/gen_cons.c:237,239 -- far
# below what two random sequences of this length reach)
FIVE, THREE = rnd(24), rnd(20)
# a second pair that OVERLAPS in the unit: THREE2 = Z + S, rc(FIVE2) = S + W (for the case where the reverse-strand placement test fails)
_S, _Z, _W = rnd(6), rnd(14), rnd(18)
THREE2, FIVE2 = _Z + _S, rc(_S + _W)
X, Y, T, T1, T2 = rnd(30), rnd(46), rnd(100), rnd(60), rnd(40)
A, B, CC, D, E = rnd(20), rnd(30), rnd(50), rnd(30), rnd(20)
M1, M2, M3 = rnd(25), rnd(40), rnd(33)
F1, RF1 = sub1(FIVE, 11), sub1(rc(FIVE), 7)

# name -> (consensus before trimming, adapter pair, expected sequence or None = untrimmed / "" = no record, expected fullLen)
ADAPTER_CASES = {
    # |--5'--|--target--|--rc(3')--|, 3' starts after 5' ends: tar = (_5_end + 1, _3_start - 1), src/gen_cons.c:250-255
    "forward_no_wrap": (X + FIVE + T + rc(THREE) + Y, 0, T, 1),
    # the first 3' hit of cons + cons lies BEFORE the 5' hit: the target wraps around the consensus end, src/gen_cons.c:242-249
    "forward_wrap": (T2 + rc(THREE) + Y + X + FIVE + T1, 0, T1 + T2, 1),
    # no 5' adapter on the forward strand (_5_ed == -1 -> goto REV, src/gen_cons.c:238): |--3'--|--target--|--rc(5')--| gives fullLen 2, :277-280
    "forward_miss_then_reverse": (X + THREE + T + rc(FIVE) + Y, 0, T, 2),
    "reverse_wrap": (T2 + rc(FIVE) + Y + X + THREE + T1, 0, T1 + T2, 2),                     # src/gen_cons.c:267-273
    # forward total edit distance 1, reverse total 1: REV needs STRICTLY smaller (src/gen_cons.c:265) -> the forward trim stays
    "reverse_tie_keeps_forward": (A + F1 + B + THREE + CC + RF1 + D + rc(THREE) + E, 0, B + THREE + CC + RF1 + D, 1),
    # forward total 1, reverse total 0: REV replaces it
    "reverse_strictly_smaller": (A + F1 + B + THREE + CC + rc(FIVE) + D + rc(THREE) + E, 0, CC, 2),
    # forward total 0: written at once (src/gen_cons.c:258), the exact reverse pair is never looked at
    "forward_exact_short_circuit": (A + FIVE + B + THREE + CC + rc(FIVE) + D + rc(THREE) + E, 0, B + THREE + CC + rc(FIVE) + D, 1),
    # reverse total 0 < forward total 1, but rc(5') straddles the consensus end (_5_end >= cons_len) while _5_start <= _3_end (the two
    # overlap by S): the wrap test of src/gen_cons.c:267 fails, NOTHING is assigned -- the forward target and fullLen 1 stay
    "reverse_smaller_but_placement_fails": (_W[5:] + M1 + sub1(FIVE2, 9) + T + rc(THREE2) + M2 + M3 + _Z + _S + _W[:5], 1, T, 1),
    # the same overlapping pair not straddling: _5_start <= _3_end, wrap test holds: tar = (_3_end + 1, _5_start + cons_len - 1) = all but S
    "reverse_overlapping_pair_wraps": (M1 + sub1(FIVE2, 9) + T + rc(THREE2) + M2 + _Z + _S + _W + M3, 1, None, 2),
    # empty target: tar_end = tar_start - 1; one-base target: tar_end == tar_start.  The trim needs tar_end > tar_start (src/gen_cons.c:282):
    # the consensus is written UNTRIMMED, and fullLen stays 1
    "empty_target_not_trimmed": (X + FIVE + rc(THREE) + Y + T, 0, None, 1),
    "one_base_target_not_trimmed": (X + FIVE + b"G" + rc(THREE) + Y + T, 0, None, 1),
    # a two-base target IS trimmed -- to 2 bases, which the length filter then drops: the filter sees the TRIMMED length (src/gen_cons.c:11,294)
    "two_base_target_trimmed_then_filtered": (X + FIVE + b"GA" + rc(THREE) + Y + T, 0, b"", 0),
    # the search runs only if cons_len > five_len + three_len (src/gen_cons.c:226): 44 bases -> no search, fullLen 0; 45 -> found, empty target
    "consensus_as_long_as_both_adapters": (FIVE + rc(THREE), 0, None, 0),
    "consensus_one_base_longer": (FIVE + rc(THREE) + b"A", 0, None, 1),
}
ADAPTER_CASES["reverse_overlapping_pair_wraps"] = (ADAPTER_CASES["reverse_overlapping_pair_wraps"][0], 1,
                                                   (lambda c: (c + c)[c.index(_Z + _S) + 20: c.index(_S + _W) + len(c)])(ADAPTER_CASES["reverse_overlapping_pair_wraps"][0]), 2)


@pytest.fixture(scope="module")
def adapter_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("kat_adapters")
    out = []
    for i, (f, t) in enumerate(((FIVE, THREE), (FIVE2, THREE2))):
        fns = (str(d / ("five%d.fa" % i)), str(d / ("three%d.fa" % i)))
        open(fns[0], "wb").write(b">five\n" + f + b"\n")
        open(fns[1], "wb").write(b">three\n" + t + b"\n")
        out.append(fns)
    return out


BACKENDS = [pytest.param("oracle"), pytest.param("emul"), pytest.param("gpu", marks=pytest.mark.gpu)]


def records(backend, reads, adapters=None, **kw):
    """tidehunter_core() on every read -> list (per read) of record dicts, through the named implementation"""
    if backend == "oracle":
        para = O.default_para(**kw)
        if adapters:
            O.load().tho_para_load_adapters(para, adapters[0].encode(), adapters[1].encode())
        return [O.core(s, para) for s in reads]
    th = TH.TideHunter(lib_path=U.emul_lib_path(), adapters=adapters, **kw) if backend == "emul" else TH.TideHunter(device=0, adapters=adapters, **kw)
    try:
        return th.process(reads)
    finally:
        th.close()


def cli(backend, tmp_path, reads, *args, names=None):
    """the command line's stdout on a FASTA file of `reads`"""
    fa = str(tmp_path / "in.fa")
    with open(fa, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">" + (names[i] if names else b"r%d" % i) + b"\n" + s + b"\n")
    if backend == "oracle":
        exe = os.path.join(ROOT, "oracle", "tho_oracle")
        if not os.path.exists(exe):
            O.build()
    elif backend == "emul":
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpu_emul"), "TideHunter-emul"], check=True, stdout=subprocess.DEVNULL)
        exe = os.path.join(ROOT, "tests", "cpu_emul", "TideHunter-emul")
    else:
        exe = os.path.join(ROOT, "tidehunter_amd", "TideHunter-amd")
    r = subprocess.run([exe, *args, fa], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    return r.stdout


def check_error_free_record(r, cons, read, k=K):
    """what every error-free read must give, whatever the adapters decide (module docstring)"""
    P = len(cons)
    n_b = (len(read) - k) // P + 1                                     # boundaries k - 1 + i P that fit the read
    assert r["sub_pos"] == [k - 1 + i * P for i in range(n_b)]
    assert r["ave_match"] == 100.0                                     # iden_n * 100 / (len + 0.0) with iden_n == len, averaged (src/gen_cons.c:208-216)
    # copyNum = n_seqs + (max_q + 1.0) / cons_len twice (src/gen_cons.c:218-223): left flank k bases, right flank what is left of the read
    left, right = k, len(read) - (k - 1 + (n_b - 1) * P) - 1
    want = float(n_b - 1)
    want += (left - 1 + 1.0) / P
    want += (right - 1 + 1.0) / P if right > 0 else 0.0                # (tlen <= 0: ksw2_right_ext leaves max_q = -1, src/ksw2_align.c:57 -> + 0 / cons_len)
    assert r["copy_num"] == want
    assert (r["start"], r["end"]) == (0, len(read) - 1)                # cons_start = par_pos[0] - max_t, cons_end = par_pos[last] + max_t + 1


@pytest.mark.parametrize("backend", BACKENDS)
def test_adapter_decision_branches(backend, adapter_files):
    """every branch of src/gen_cons.c:226-291; ALL cases go through one batch (one handle), each record checked against its typed answer"""
    for pair in (0, 1):
        names = [n for n, c in ADAPTER_CASES.items() if c[1] == pair]
        reads = [read_of(ADAPTER_CASES[n][0]) for n in names]
        got = records(backend, reads, adapters=adapter_files[pair])
        for n, read, recs in zip(names, reads, got):
            cons, _, want_seq, want_fl = ADAPTER_CASES[n]
            if want_seq == b"":
                assert recs == [], n
                continue
            assert len(recs) == 1, n
            r = recs[0]
            want = cons if want_seq is None else want_seq
            assert r["seq"] == want, (n, r["seq"], want)
            assert r["cons_len"] == len(want) and r["full_length"] == want_fl, (n, r["cons_len"], r["full_length"])
            check_error_free_record(r, cons, read)                     # start / end / copyNum use the UNTRIMMED consensus (src/gen_cons.c:218-223,294)


@pytest.mark.parametrize("backend", BACKENDS)
def test_filters_see_the_trimmed_consensus(backend, adapter_files):
    """min_len is compared with the length AFTER adapter trimming (src/gen_cons.c:11 called from :294); -F drops fullLen 0 (:293)"""
    cons = X + FIVE + T + rc(THREE) + Y                                # 220 bases before trimming, 100 after
    read = read_of(cons)
    assert [len(r) for r in records(backend, [read, read], adapters=adapter_files[0], min_len=100)] == [1, 1]
    assert records(backend, [read], adapters=adapter_files[0], min_len=101) == [[]]
    plain = rnd(220, seeded(101))                                      # no adapter anywhere: fullLen 0
    got = records(backend, [read_of(plain), read], adapters=adapter_files[0], only_full_length=1)
    assert got[0] == [] and len(got[1]) == 1 and got[1][0]["full_length"] == 1
    got = records(backend, [read_of(plain)], adapters=adapter_files[0])
    assert got[0][0]["full_length"] == 0 and got[0][0]["seq"] == plain


@pytest.mark.parametrize("backend", BACKENDS)
def test_only_longest_keeps_the_longer_span(backend):
    """-l (src/gen_cons.c:12-16): the second consensus replaces the first iff its span end - start is STRICTLY larger; else it is dropped"""
    g = seeded(102)
    ua, ub = rnd(96, g), rnd(150, g)
    short_then_long = read_of(ua, 6) + read_of(ub, 10)                 # spans ~576 and ~1500 bases
    long_then_short = read_of(ua, 20) + read_of(ub, 6)                 # spans ~1920 and ~900 bases
    both = records(backend, [short_then_long])[0]
    # chains by ascending end (src/tandem_chain.c:393-396).  The first repeat starts the read, so its consensus is ua exactly; where the second one's first boundary falls
    # depends on how many bases of the junction happen to continue its first k-mer backwards: its consensus is ub up to a rotation
    assert len(both) == 2 and both[0]["seq"] == ua and is_rotation(both[1]["seq"], ub)
    got = records(backend, [short_then_long, long_then_short], only_longest=1)
    assert len(got[0]) == 1 and is_rotation(got[0][0]["seq"], ub) and [r["seq"] for r in got[1]] == [ua]
    assert got[0][0]["end"] - got[0][0]["start"] > both[0]["end"] - both[0]["start"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_two_units_consensus_is_the_first_unit(backend):
    """n_seqs <= 2 (src/abpoa_cons.c:57-83): no MSA; the consensus is unit 0 VERBATIM even when unit 1 differs; with a coverage bound the two
    units must be identical (_min_cov 2) or the run is dropped; FASTQ qualities are '!' (33)"""
    P2 = rnd(256, seeded(103))
    u = P2[-K:] + P2[:-K]
    same = u * 2 + u[:60]                                              # boundaries 7, 263, 519: two units
    differ = u + sub1(u, 100) + u[:60]
    r = records(backend, [same, differ])
    for recs, read in zip(r, (same, differ)):
        assert len(recs) == 1 and recs[0]["seq"] == P2 and recs[0]["sub_pos"] == [7, 263, 519]
        assert recs[0]["copy_num"] == 2 + 8 / 256 + 52 / 256           # 2.234375, exact in binary
    assert r[0][0]["ave_match"] == 100.0
    assert r[1][0]["ave_match"] == (256 * 100 / 256.0 + 255 * 100 / 256.0) / 2      # 99.8046875: the one substitution costs unit 1 one identity
    q = records(backend, [same], out_fmt=3)[0][0]
    assert q["qual"] == b"!" * 256                                     # src/abpoa_cons.c:79
    assert len(records(backend, [same], min_cov=2)[0]) == 1            # identical units: _min_cov = 2 >= 2
    assert len(records(backend, [differ], min_cov=1)[0]) == 1          # _min_cov = 1 >= 1
    # (differ with min_cov = 2: the reference never returns -- cons_len == 0 `continue`s without advancing i, src/gen_cons.c:206;
    # this repository skips the run instead: DESIGN.md section 2)
    assert records(backend, [differ], min_cov=2) == [[]]
    assert len(records(backend, [same], min_frac=0.9)[0]) == 1         # (int)(2 * 0.9) = 1 (src/abpoa_cons.c:55)


def _unit256():
    P2 = rnd(256, seeded(104))
    return P2, P2[-K:] + P2[:-K]


@pytest.mark.parametrize("backend", BACKENDS)
def test_output_line_formatting(backend, tmp_path):
    """mini_tandem_output (src/main.c:237-263): `>name_repN_%.1f len_start+1_end+1_consLen_%.1f_fullLen_subPos`, subPos + 2 for all but the last
    (+ 1); %.1f of a value that is EXACTLY halfway rounds to even (glibc printf): 8.25 -> 8.2, 8.75 -> 8.8"""
    P2, u = _unit256()
    reads = [u * 8 + u[:64], u * 8 + u[:192], u * 8 + u[:100]]         # copyNum 8 + 8/256 + {56, 184, 92}/256 = 8.25, 8.75, 8.390625
    sub = "9,265,521,777,1033,1289,1545,1801,2056"                     # 7 + 256 i, + 2; the last one 2055 + 1
    want = b"".join(b">%s_rep0_%s %d_1_%d_256_100.0_0_%s\n%s\n" % (nm, cn, len(s), len(s), sub.encode(), P2)
                    for nm, cn, s in zip((b"readA", b"readB", b"readC"), (b"8.2", b"8.8", b"8.4"), reads))
    assert cli(backend, tmp_path, reads, names=[b"readA", b"readB", b"readC"]) == want
    # tabular (-f 2, src/main.c:243-247): name rep copyNum readLen start end consLen aveMatch fullLen subPos cons
    want_tab = b"".join(b"%s\trep0\t%s\t%d\t1\t%d\t256\t100.0\t0\t%s\t%s\n" % (nm, cn, len(s), len(s), sub.encode(), P2)
                        for nm, cn, s in zip((b"readA", b"readB", b"readC"), (b"8.2", b"8.8", b"8.4"), reads))
    assert cli(backend, tmp_path, reads, "-f", "2", names=[b"readA", b"readB", b"readC"]) == want_tab
    # FASTQ (-f 3, src/main.c:248-262): '@' header, '+' line, Phred of full coverage: x = 13.8 * (1.25 * 8 / 8 - 0.25) = 13.8,
    # p = 1 - 1 / (1 + e^-13.8), 33 + (int)(-10 log10 p + 0.499) = 33 + 60 = ']' (src/abpoa_cons.c:99-107)
    want_fq = b"@readA_rep0_8.2 2112_1_2112_256_100.0_0_%s\n%s\n+\n%s\n" % (sub.encode(), P2, b"]" * 256)
    assert cli(backend, tmp_path, reads[:1], "-f", "3", names=[b"readA"]) == want_fq
    # unit mode (-u, src/main.c:220-235): FASTA prints bases sub_pos[i] + 1 .. sub_pos[i + 1]; the TAB form stops ONE BASE EARLY (`j <` at :231)
    units = cli(backend, tmp_path, reads[:1], "-u", names=[b"readA"]).split(b"\n")
    assert units[0] == b">readA_rep0_sub0" and units[1] == P2 and units[14] == b">readA_rep0_sub7" and len(units) == 17
    tab_units = cli(backend, tmp_path, reads[:1], "-u", "-f", "2", names=[b"readA"]).split(b"\n")
    assert tab_units[0] == b"readA\trep0\tsub0\t" + P2[:-1] and len(tab_units) == 9


@pytest.mark.parametrize("backend", BACKENDS)
def test_option_parsing_quirks(backend, tmp_path, adapter_files):
    """main(), src/main.c:54-63,438-535"""
    P2, u = _unit256()
    read = u * 8
    out = cli(backend, tmp_path, [read])
    # eight units need nine boundaries 7 .. 2055 but the read ends at 2047: seven units, boundaries 7 .. 1799, copyNum 7 + 8/256 + 248/256 = 8.0
    assert out == b">r0_rep0_8.0 2048_1_2048_256_100.0_0_9,265,521,777,1033,1289,1545,1800\n" + P2 + b"\n"
    # K / M / G suffixes on -p / -P only (th_parse_num, src/main.c:54-63): -P 0.2K = 200 < period 256 -> the hits are gone; -P 0.3K = 300 keeps them
    assert cli(backend, tmp_path, [read], "-P", "0.2K") == b""
    assert cli(backend, tmp_path, [read], "-P", "0.3k") == out
    # -p 0.3K = 300 > 256: the walk back over a k-mer's earlier occurrences stops at the first distance >= min_p (src/tandem_hit.c:192-196), which is
    # TWO units: period 512, boundaries 7 + 512 i, three units of P2 + P2, copyNum 3 + 8/512 + 504/512 = 4.0
    assert cli(backend, tmp_path, [read], "-p", "0.3K") == b">r0_rep0_4.0 2048_1_2048_512_100.0_0_9,521,1033,1544\n" + P2 + P2 + b"\n"
    # -m is atoi (src/main.c:488): "1K" is 1, not 1000 -- the 256-base consensus passes
    assert cli(backend, tmp_path, [read], "-m", "1K") == out
    assert cli(backend, tmp_path, [read], "-m", "257") == b""
    # -r below 1 is a fraction of the units, from 1 on a count rounded at .499 (src/main.c:489-492): 7 identical units cover every column 7 times
    assert cli(backend, tmp_path, [read], "-r", "0.99") == out         # (int)(7 * 0.99) = 6 <= 7
    assert cli(backend, tmp_path, [read], "-r", "7.4") == out          # (int)(7.4 + .499) = 7
    assert cli(backend, tmp_path, [read], "-r", "7.6") == b""          # 8 > 7: dropped (src/abpoa_cons.c:94-98)
    # long names with underscores (src/main.c:29-30); -O / -E take "first[,second]" (src/main.c:478-479)
    assert cli(backend, tmp_path, [read], "--gap_open", "6,30", "--gap_ext", "3", "--match", "3", "--mismatch", "5") == out
    # -c below MIN_COPY = 2 ends the run without output (src/main.c:462-466); 'q' is in the optstring without a case -> default: End (:441,519)
    assert cli(backend, tmp_path, [read], "-c", "1") == b""
    assert cli(backend, tmp_path, [read], "-q") == b""
    assert cli(backend, tmp_path, [read], "-f", "5") == b""            # unknown format number (src/main.c:509-513)
    assert cli(backend, tmp_path, [read], "-u", "-f", "3") == b""      # units only as FASTA / TAB (src/main.c:522-525)
    assert cli(backend, tmp_path, [read], "-F") == b""                 # -F needs both adapters (src/main.c:530-533)
    assert cli(backend, tmp_path, [read], "-c", "8") == b""            # j - i > min_copy (src/gen_cons.c:199): 8 boundaries are not > 8
    assert cli(backend, tmp_path, [read], "-c", "7") == out
    # -a: the thresholds are (int)(len * (1 - rat)) with a FLOAT rat (src/gen_cons.c:237): one substitution in the 5' adapter passes at the default
    # 0.8 (k = 4) and fails at 0.99 (k = 0: exact only), where the record keeps fullLen 0 and its untrimmed consensus
    cons = X + F1 + T + rc(THREE) + Y
    rd = read_of(cons)
    hdr = lambda fl, n: b">r0_rep0_8.0 1760_1_1760_%d_100.0_%d_9,229,449,669,889,1109,1329,1548\n" % (n, fl)
    assert cli(backend, tmp_path, [rd], "-5", adapter_files[0][0], "-3", adapter_files[0][1]) == hdr(1, 100) + T + b"\n"
    assert cli(backend, tmp_path, [rd], "-5", adapter_files[0][0], "-3", adapter_files[0][1], "-a", "0.99") == hdr(0, 220) + cons + b"\n"
