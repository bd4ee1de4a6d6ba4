"""Pins the oracle's ksw2 / edlib / seq restatements against oracle/_ref, i.e. against the
reference's OWN ksw2/ksw2_extz2_sse.c, src/ksw2_align.c, edlib/src/edlib.cpp, src/edlib_align.c and
src/seq.c compiled where they lie (oracle/Makefile).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def _mutate(rng, s, rate):
    out = []
    for b in s:
        r = rng.random()
        if r < rate / 3:
            out.append(int(rng.integers(0, 4)))
        elif r < 2 * rate / 3:
            out.append(int(b)); out.append(int(rng.integers(0, 4)))
        elif r < rate:
            continue
        else:
            out.append(int(b))
    return np.array(out if out else [0], dtype=np.uint8)


def _pairs(seed, n, maxlen):
    rng = np.random.default_rng(seed)
    for it in range(n):
        kind = it % 6
        la = int(rng.integers(1, maxlen))
        if kind == 0:      # unrelated
            a = rng.integers(0, 4, la); b = rng.integers(0, 4, int(rng.integers(1, maxlen)))
        elif kind == 1:    # related, noisy
            a = rng.integers(0, 4, la); b = _mutate(rng, a, 0.15)
        elif kind == 2:    # low complexity: tie-prone
            a = rng.integers(0, 2, la); b = _mutate(rng, a, 0.2)
        elif kind == 3:    # homopolymers
            a = np.repeat(rng.integers(0, 4, max(1, la // 5)), 5); b = _mutate(rng, a, 0.1)
        elif kind == 4:    # with N (code 4) and '-' (code 5)
            a = rng.integers(0, 4, la); b = _mutate(rng, a, 0.1)
            a = a.copy(); a[rng.random(len(a)) < 0.05] = 4
            b = b.copy(); b[rng.random(len(b)) < 0.05] = 4
            if len(a) > 3: a[len(a) // 2] = 5
            if len(b) > 3: b[len(b) // 3] = 5
        else:              # very short
            a = rng.integers(0, 4, int(rng.integers(1, 4))); b = rng.integers(0, 4, int(rng.integers(1, 6)))
        yield np.asarray(a, dtype=np.uint8), np.asarray(b, dtype=np.uint8)


def test_ksw2_global_cigar_matches_reference(refso):
    n_checked = 0
    for q, t in _pairs(11, 600, 160):
        qa, qp = O.u8(q); ta, tp = O.u8(t)
        n = C.c_int(); cg = C.POINTER(C.c_uint32)()
        iden_ref = refso.ksw2_global_with_cigar(qp, len(qa), tp, len(ta), C.byref(n), C.byref(cg))
        ref_cigar = [cg[i] for i in range(n.value)]
        iden, cigar, _ = O.ksw2_global_cigar(q, t)
        assert cigar == ref_cigar, (q.tolist(), t.tolist())
        assert iden == iden_ref
        # back-projection for every admissible q_left_ext on a few pairs
        if n_checked % 10 == 0:
            arr = (C.c_uint32 * len(cigar))(*cigar)
            for ext in range(0, len(q) + 1, max(1, len(q) // 7)):
                a = refso.ksw2_backtrack_left_end(n.value, cg, len(q), len(t), ext)
                b = O.load().tho_ksw2_backtrack_left_end(len(cigar), arr, len(q), len(t), ext)
                assert a == b
        n_checked += 1
    assert n_checked == 600


def test_ksw2_global_long_matches_reference(refso):
    for q, t in _pairs(5, 12, 1400):
        qa, qp = O.u8(q); ta, tp = O.u8(t)
        assert O.load().tho_ksw2_global(qp, len(qa), tp, len(ta)) == refso.ksw2_global(qp, len(qa), tp, len(ta))


@pytest.mark.parametrize("left", [False, True])
def test_ksw2_extension_matches_reference(refso, left):
    for q, t in _pairs(23 + left, 600, 120):
        qa, qp = O.u8(q); ta, tp = O.u8(t)
        mq, mt = C.c_int(-9), C.c_int(-9)
        (refso.ksw2_left_ext if left else refso.ksw2_right_ext)(qp, len(qa), tp, len(ta), C.byref(mq), C.byref(mt))
        assert O.ksw2_ext(q, t, left) == (mq.value, mt.value), (q.tolist(), t.tolist())


def test_ksw2_extension_unit_vs_flank_shapes(refso):
    """shape used by src/gen_cons.c:218-223: consensus (query) against a longer or empty-ish flank"""
    rng = np.random.default_rng(3)
    for it in range(40):
        cons = rng.integers(0, 4, int(rng.integers(30, 400))).astype(np.uint8)
        flank = np.concatenate([_mutate(rng, cons[: int(rng.integers(1, len(cons)))], 0.1),
                                rng.integers(0, 4, int(rng.integers(0, 600))).astype(np.uint8)])
        qa, qp = O.u8(cons); ta, tp = O.u8(flank)
        mq, mt = C.c_int(), C.c_int()
        refso.ksw2_right_ext(qp, len(qa), tp, len(ta), C.byref(mq), C.byref(mt))
        assert O.ksw2_ext(cons, flank) == (mq.value, mt.value)


def test_edlib_hw_matches_reference(refso):
    rng = np.random.default_rng(7)
    alpha = b"ACGTNacgtn"
    n = 0
    for it in range(1500):
        ql = int(rng.integers(1, 140)) if it % 3 else int(rng.choice([25, 64, 84, 128]))
        q = bytes(alpha[i] for i in rng.integers(0, 4, ql))
        tl = int(rng.integers(1, 500))
        t = bytearray(alpha[i] for i in rng.integers(0, 4, tl))
        if it % 2 == 0 and tl > ql + 2:      # plant a noisy copy
            pos = int(rng.integers(0, tl - ql))
            cp = bytearray(q)
            for _ in range(int(rng.integers(0, max(1, ql // 5)))):
                cp[int(rng.integers(0, ql))] = alpha[int(rng.integers(0, 10))]
            t[pos:pos + ql] = cp
        if it % 5 == 0:
            t = bytearray(bytes(t).lower())
        k = int(ql * (1 - np.float32(0.8))) if it % 4 else int(rng.integers(0, ql + 3))
        s, e = C.c_int(-7), C.c_int(-7)
        ed = refso.edlib_align_HW(q, ql, bytes(t), len(t), C.byref(s), C.byref(e), k)
        got = O.edlib_hw(q, bytes(t), k)
        assert got[0] == ed, (q, bytes(t), k)
        if ed >= 0:
            assert got == (ed, s.value, e.value), (q, bytes(t), k)
            n += 1
    assert n > 300


def test_adapter_k_is_float_truncation():
    """src/gen_cons.c:240: k = len * (1 - ada) evaluated in float: 84 -> 16, 25 -> 4"""
    assert int(np.float32(84) * (np.float32(1) - np.float32(0.8))) == 16
    assert int(np.float32(25) * (np.float32(1) - np.float32(0.8))) == 4


def test_encode_and_revcomp_match_reference(refso):
    data = bytes(range(1, 128)) + b"ACGTNacgtn-RYKM\x00\x01\x02\x03"
    ref = refso.get_bseq(data, len(data))
    got = O.encode(data)
    assert got.tolist() == [ref[i] for i in range(len(data))]
    lib = O.load()
    a = C.string_at(refso.get_rc_seq(data, len(data)), len(data))
    b = C.string_at(lib.tho_rc_seq(data, len(data)), len(data))
    assert a == b
