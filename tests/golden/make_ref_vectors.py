#!/usr/bin/env python3
"""Writes tests/golden/ref_ksw2_vectors.npz and ref_edlib_vectors.npz: input / output vectors of the REFERENCE's own alignment code.

Run in the build container (needs /root/reference): the functions called are the reference's, compiled where they lie into
oracle/_ref/libthref.so by oracle/Makefile -- ksw2/ksw2_extz2_sse.c + src/ksw2_align.c (ksw2_global, ksw2_global_with_cigar,
ksw2_backtrack_left_end, ksw2_left_ext, ksw2_right_ext: src/ksw2_align.c:88-173) and edlib/src/edlib.cpp + src/edlib_align.c
(edlib_align_HW: src/edlib_align.c:73-85).  The files hold DATA only (sequences as code / byte arrays and the integers the reference
returned); nothing of the reference's source travels.  Consumers: tests/test_ref_vectors.py (the oracle, CPU, any box) and
tests/test_gpu_ref_vectors.py (th_k_align / th_ksw_global_wave / th_k_adapter* through the raw-pair hooks, on the GPU box, where
/root/reference does not exist).

    python tests/golden/make_ref_vectors.py          # deterministic: the same files every time
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

N_EXT = 4


def mutate(rng, s, rate):
    out = []
    for b in s:
        r = rng.random()
        if r < rate / 3:
            out.append((int(b) + 1 + int(rng.integers(0, 3))) & 3 if b < 4 else int(rng.integers(0, 4)))
        elif r < 2 * rate / 3:
            out.append(int(b)); out.append(int(rng.integers(0, 4)))
        elif r < rate:
            continue
        else:
            out.append(int(b))
    return np.array(out if out else [0], dtype=np.uint8)


def ksw2_pairs():
    """(class name, q, t) -- q plays the unit / the consensus, t the consensus / the flank"""
    rng = np.random.default_rng(20250104)
    # short pairs: the kinds of tests/test_oracle_ref.py and the degenerate ones
    for it in range(2200):
        kind = it % 8
        la = int(rng.integers(1, 200))
        if kind == 0:      # unrelated
            a = rng.integers(0, 4, la); b = rng.integers(0, 4, int(rng.integers(1, 200)))
        elif kind == 1:    # related, noisy
            a = rng.integers(0, 4, la); b = mutate(rng, a, 0.15)
        elif kind == 2:    # low complexity: tie-prone
            a = rng.integers(0, 2, la); b = mutate(rng, a, 0.2)
        elif kind == 3:    # homopolymer runs
            a = np.repeat(rng.integers(0, 4, max(1, la // 5)), 5); b = mutate(rng, a, 0.1)
        elif kind == 4:    # with N (code 4)
            a = rng.integers(0, 4, la).astype(np.uint8); b = mutate(rng, a, 0.1)
            a[rng.random(len(a)) < 0.06] = 4; b = b.copy(); b[rng.random(len(b)) < 0.06] = 4
        elif kind == 5:    # very short, incl. length 1
            a = rng.integers(0, 4, int(rng.integers(1, 4))); b = rng.integers(0, 4, int(rng.integers(1, 6)))
        elif kind == 6:    # dinucleotide repeats: every shift by two ties
            a = np.tile(rng.integers(0, 4, 2), max(1, la // 2)); b = mutate(rng, a, 0.08)
        else:              # flank shape: consensus against a longer or much shorter flank (src/gen_cons.c:218-223)
            a = rng.integers(0, 4, max(8, la)); b = np.concatenate([mutate(rng, a[: int(rng.integers(1, len(a)))], 0.1), rng.integers(0, 4, int(rng.integers(0, 260)))])
        yield "short", np.asarray(a, dtype=np.uint8), np.asarray(b, dtype=np.uint8)
    # '-' (code 5, src/seq.c:15-32): a handful, as tests/test_oracle_ref.py has them
    for it in range(24):
        a = rng.integers(0, 4, int(rng.integers(8, 120))).astype(np.uint8); b = mutate(rng, a, 0.1)
        a[len(a) // 2] = 5; b[len(b) // 3] = 5
        yield "short", a, b
    # one insertion or deletion run across a 64-column strip boundary, perfect matches after it (th_ksw.h's liveness rule)
    for it in range(240):
        kb = int(rng.integers(1, 6)); pre = kb * 64 + int(rng.integers(-4, 5)); glen = int(rng.integers(1, 9)); suf = int(rng.integers(6, 330))
        base = rng.integers(0, 4, pre + suf).astype(np.uint8)
        q, t = base.copy(), base.copy()
        for _ in range(int(rng.integers(0, 4))):
            x = int(rng.integers(0, pre - 1)); q[x] = (q[x] + 1 + int(rng.integers(0, 3))) & 3
        ins = rng.integers(0, 4, glen).astype(np.uint8)
        if it & 1:
            q = np.concatenate([q[:pre], ins, q[pre:]])
        else:
            t = np.concatenate([t[:pre], ins, t[pre:]])
        yield "strip", q, t
    # unit against consensus at the headline shape and its neighbours: 10 % and 15 % divergence
    for it in range(300):
        L = int(rng.choice([500, 1000, 1000, 1000, 2000]))
        cons = rng.integers(0, 4, L).astype(np.uint8)
        unit = mutate(rng, cons, 0.10 if it % 2 == 0 else 0.15)
        if it % 25 == 0:
            unit[rng.random(len(unit)) < 0.01] = 4
        yield "kb", unit, cons
    # flanks at that shape: the consensus against a flank that continues the repeat for a while
    for it in range(60):
        cons = rng.integers(0, 4, 1000).astype(np.uint8)
        keep = int(rng.integers(1, 1000))
        flank = np.concatenate([mutate(rng, cons[:keep], 0.10 if it % 2 else 0.15), rng.integers(0, 4, int(rng.integers(0, 1500))).astype(np.uint8)])
        yield "kb", cons, flank
    # long periods (config 4): 5000-base pairs at 10 % and 15 %
    for it in range(24):
        cons = rng.integers(0, 4, 5000 if it % 3 else 3000).astype(np.uint8)
        yield "long", mutate(rng, cons, 0.10 if it % 2 == 0 else 0.15), cons
    for it in range(8):
        cons = rng.integers(0, 4, 5000).astype(np.uint8)
        keep = int(rng.integers(500, 5000))
        yield "long", cons, np.concatenate([mutate(rng, cons[:keep], 0.15), rng.integers(0, 4, int(rng.integers(0, 3000))).astype(np.uint8)])


def make_ksw2(ref):
    cls_names = ["short", "strip", "kb", "long"]
    qs, ts, cls, iden, lext, rext, xs, tbs = [], [], [], [], [], [], [], []
    for name, q, t in ksw2_pairs():
        qa, qp = O.u8(q); ta, tp = O.u8(t)
        n = C.c_int(); cg = C.POINTER(C.c_uint32)()
        v = ref.ksw2_global_with_cigar(qp, len(qa), tp, len(ta), C.byref(n), C.byref(cg))
        assert v == ref.ksw2_global(qp, len(qa), tp, len(ta))
        ql = len(qa)
        x = sorted({1, max(1, ql // 3), max(1, ql // 2), ql})
        x = (x + [ql] * N_EXT)[:N_EXT]
        tb = [ref.ksw2_backtrack_left_end(n.value, cg, ql, len(ta), e) for e in x]
        mq, mt = C.c_int(-9), C.c_int(-9)
        ref.ksw2_left_ext(qp, ql, tp, len(ta), C.byref(mq), C.byref(mt)); le = (mq.value, mt.value)
        ref.ksw2_right_ext(qp, ql, tp, len(ta), C.byref(mq), C.byref(mt)); re_ = (mq.value, mt.value)
        qs.append(qa); ts.append(ta); cls.append(cls_names.index(name)); iden.append(v); lext.append(le); rext.append(re_); xs.append(x); tbs.append(tb)
    q_off = np.concatenate([[0], np.cumsum([len(a) for a in qs])]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum([len(a) for a in ts])]).astype(np.int32)
    out = os.path.join(HERE, "ref_ksw2_vectors.npz")
    np.savez_compressed(out, q_cat=np.concatenate(qs), q_off=q_off, t_cat=np.concatenate(ts), t_off=t_off, cls=np.array(cls, dtype=np.int8),
                        iden=np.array(iden, dtype=np.int32), left_ext=np.array(lext, dtype=np.int32), right_ext=np.array(rext, dtype=np.int32),
                        q_left_ext=np.array(xs, dtype=np.int32), t_back=np.array(tbs, dtype=np.int32), cls_names=np.array(cls_names))
    print("%s: %d pairs (%s), %.1f KB" % (out, len(qs), ", ".join("%d %s" % (cls.count(i), c) for i, c in enumerate(cls_names)), os.path.getsize(out) / 1e3))


def rand_text(rng, n, alpha):
    return bytes(alpha[i] for i in rng.integers(0, len(alpha), n))


def make_edlib(ref):
    rng = np.random.default_rng(77)
    up, mixed = b"ACGT", b"ACGTacgtNn"
    # queries in groups of four (the device searches four per job): adapter-like lengths and the 64-symbol block boundaries
    qlens = [84, 25, 84, 25, 1, 5, 63, 64, 65, 128, 130, 200, 256, 33, 100, 12]
    queries = [rand_text(rng, L, up if i % 3 else mixed) for i, L in enumerate(qlens)]
    ks = []
    for i, L in enumerate(qlens):
        if i in (4, 5):
            ks.append(int(rng.integers(0, 3)))
        elif i == 13:
            ks.append(-1)                                     # edlib: no limit
        elif i == 14:
            ks.append(L + 2)                                  # above the query length
        elif i == 15:
            ks.append(0)                                      # exact occurrences only
        else:
            ks.append(int(np.float32(L) * (np.float32(1) - np.float32(0.8))))       # src/gen_cons.c:240: evaluated in float (84 -> 16, 25 -> 4)
    # byte targets (the -s path searches whole reads as uploaded): planted noisy copies, both cases, N, other bytes
    targets = []
    for it in range(360):
        tl = int(rng.integers(1, 3000)) if it % 9 else int(rng.integers(1, 40))
        t = bytearray(rand_text(rng, tl, mixed if it % 4 == 0 else up))
        for _ in range(int(rng.integers(0, 4))):
            q = queries[int(rng.integers(0, len(queries)))]
            if tl > len(q) + 2:
                pos = int(rng.integers(0, tl - len(q)))
                cp = bytearray(q)
                for _e in range(int(rng.integers(0, max(1, len(q) // 5)))):
                    cp[int(rng.integers(0, len(q)))] = mixed[int(rng.integers(0, len(mixed)))]
                if rng.random() < 0.3 and len(cp) > 4:
                    del cp[int(rng.integers(0, len(cp)))]
                t[pos:pos + len(cp)] = cp
        if it % 7 == 0:
            t = bytearray(bytes(t).lower())
        if it % 31 == 0 and tl > 3:
            t[tl // 2] = ord("-")
        targets.append(bytes(t))
    res = np.zeros((len(targets), len(queries), 3), dtype=np.int32)
    for i, t in enumerate(targets):
        for j, q in enumerate(queries):
            s, e = C.c_int(-1), C.c_int(-1)
            ed = ref.edlib_align_HW(q, len(q), t, len(t), C.byref(s), C.byref(e), ks[j])
            res[i, j] = (ed, s.value, e.value) if ed != -1 else (-1, -1, -1)
    # consensus targets (codes 0..4), searched over consensus + consensus as seqs_msa() does (src/gen_cons.c:228-231)
    cons, res2 = [], []
    for it in range(260):
        cl = int(rng.integers(2, 2600)) if it % 8 else int(rng.integers(2, 60))
        c = rng.integers(0, 4, cl).astype(np.uint8)
        if it % 10 == 0:
            c[rng.random(cl) < 0.02] = 4
        for _ in range(int(rng.integers(0, 3))):
            q = queries[int(rng.integers(0, len(queries)))].upper()
            qc = np.array([b"ACGTN".index(ch) for ch in q], dtype=np.uint8)
            if cl > len(qc) + 2:
                pos = int(rng.integers(0, cl))                # may wrap around the end: the doubled text holds it
                for x in range(len(qc)):
                    if rng.random() > 0.08:
                        c[(pos + x) % cl] = qc[x]
        cons.append(c)
        txt = bytes(b"ACGTN"[v] for v in c) * 2
        row = []
        for j, q in enumerate(queries):
            s, e = C.c_int(-1), C.c_int(-1)
            ed = ref.edlib_align_HW(q, len(q), txt, len(txt), C.byref(s), C.byref(e), ks[j])
            row.append((ed, s.value, e.value) if ed != -1 else (-1, -1, -1))
        res2.append(row)
    out = os.path.join(HERE, "ref_edlib_vectors.npz")
    np.savez_compressed(out, queries=np.frombuffer(b"".join(queries), dtype=np.uint8), q_off=np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32), k=np.array(ks, dtype=np.int32),
                        t_cat=np.frombuffer(b"".join(targets), dtype=np.uint8), t_off=np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.int32), res=res,
                        c_cat=np.concatenate(cons), c_off=np.concatenate([[0], np.cumsum([len(c) for c in cons])]).astype(np.int32), res_doubled=np.array(res2, dtype=np.int32))
    print("%s: %d queries x (%d byte targets + %d doubled consensuses), %.1f KB; hits: %d / %d" % (out, len(queries), len(targets), len(cons), os.path.getsize(out) / 1e3,
                                                                                                int((res[:, :, 0] >= 0).sum()), int((np.array(res2)[:, :, 0] >= 0).sum())))


if __name__ == "__main__":
    ref = O.load_ref()
    if ref is None:
        sys.exit("oracle/_ref/libthref.so is not built (needs /root/reference): run `make -C oracle ref` in the build container")
    make_ksw2(ref)
    make_edlib(ref)
