"""ctypes access to the CPU oracle (oracle/libtho.so) and to oracle/_ref -- TEST-ONLY helpers."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class Para(C.Structure):
    _fields_ = [("k", C.c_int), ("w", C.c_int), ("hpc", C.c_int), ("min_copy", C.c_int), ("min_cov", C.c_int),
                ("max_div", C.c_double), ("min_frac", C.c_double), ("min_p", C.c_int64), ("max_p", C.c_int64),
                ("match", C.c_int), ("mismatch", C.c_int), ("gap_open1", C.c_int), ("gap_open2", C.c_int),
                ("gap_ext1", C.c_int), ("gap_ext2", C.c_int), ("ada_match_rat", C.c_float),
                ("five_seq", C.c_char_p), ("five_rc_seq", C.c_char_p), ("five_len", C.c_int),
                ("three_seq", C.c_char_p), ("three_rc_seq", C.c_char_p), ("three_len", C.c_int),
                ("out_fmt", C.c_int), ("min_len", C.c_int), ("only_unit", C.c_int), ("only_longest", C.c_int),
                ("only_full_length", C.c_int), ("single_copy", C.c_int), ("n_thread", C.c_int)]


class Cons(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("qual", C.c_char_p), ("cons_len", C.c_int), ("start", C.c_int), ("end", C.c_int),
                ("copy_num", C.c_double), ("ave_match", C.c_double), ("full_length", C.c_int),
                ("sub_pos", C.POINTER(C.c_int)), ("pos_n", C.c_int),
                ("n_seqs", C.c_int), ("max_q_l", C.c_int), ("max_t_l", C.c_int), ("max_q_r", C.c_int), ("max_t_r", C.c_int)]


class ReadRes(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("a", C.POINTER(Cons))]


class Cell(C.Structure):
    _fields_ = [("i", C.c_int), ("j", C.c_int)]


class Dp(C.Structure):
    _fields_ = [("from_i", C.c_int), ("from_j", C.c_int), ("start", C.c_int), ("end", C.c_int), ("score", C.c_int),
                ("is_tracked", C.c_int8)]


class Chain(C.Structure):
    _fields_ = [("cell", C.POINTER(Cell)), ("len", C.c_int), ("score", C.c_int), ("est_ch_i", C.c_int),
                ("est_period", C.c_int), ("est_start", C.c_int)]


class ChainRes(C.Structure):
    _fields_ = [("tot_n", C.c_int), ("row_size", C.POINTER(C.c_int)), ("dp", C.POINTER(C.POINTER(Dp))),
                ("ch_n", C.c_int), ("chain", C.POINTER(Chain))]


def build():
    subprocess.run(["make", "-s", "-C", ODIR, "all"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ODIR, "libtho.so")
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        build()
    lib = C.CDLL(so)
    u8p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    lib.tho_para_init.argtypes = [C.POINTER(Para)]
    lib.tho_encode.argtypes = [C.c_char_p, C.c_int, u8p]
    lib.tho_collect_hits.argtypes = [u8p, C.c_int, C.POINTER(Para), C.POINTER(C.POINTER(C.c_uint64))]
    lib.tho_chain.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.POINTER(Para), C.POINTER(ChainRes)]
    lib.tho_chain_free.argtypes = [C.POINTER(ChainRes)]
    lib.tho_partition.argtypes = [u8p, C.c_int, C.POINTER(ChainRes), C.POINTER(Chain), C.POINTER(Para), ip]
    lib.tho_partition.restype = ip
    lib.tho_ksw2_global_cigar.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, C.POINTER(C.POINTER(C.c_uint32)), ip]
    lib.tho_ksw2_global.argtypes = [u8p, C.c_int, u8p, C.c_int]
    lib.tho_ksw2_backtrack_left_end.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
    lib.tho_ksw2_right_ext.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, ip]
    lib.tho_ksw2_left_ext.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, ip]
    lib.tho_edlib_hw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, ip, ip, C.c_int]
    lib.tho_poa_consensus.argtypes = [C.c_int, C.POINTER(u8p), ip, C.POINTER(Para), u8p, ip]
    lib.tho_poa_exact.argtypes = [C.c_int, C.POINTER(u8p), ip, C.POINTER(Para), C.c_int, u8p, ip]
    lib.tho_poa_set_opt.argtypes = [C.c_int, C.c_int]
    lib.tho_core.argtypes = [C.c_char_p, C.c_int, C.POINTER(Para), C.POINTER(ReadRes)]
    lib.tho_read_res_free.argtypes = [C.POINTER(ReadRes)]
    lib.tho_para_load_adapters.argtypes = [C.POINTER(Para), C.c_char_p, C.c_char_p]
    lib.tho_rc_seq.argtypes = [C.c_char_p, C.c_int]
    lib.tho_rc_seq.restype = C.c_void_p
    lib.free = C.CDLL(None).free
    lib.free.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def load_ref():
    so = os.path.join(ODIR, "_ref", "libthref.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/ksw2"):
            subprocess.run(["make", "-s", "-C", ODIR, "ref"], check=True, stdout=subprocess.DEVNULL)
        else:
            return None
    lib = C.CDLL(so)
    u8p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    lib.ksw2_global_with_cigar.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, C.POINTER(C.POINTER(C.c_uint32))]
    lib.ksw2_global.argtypes = [u8p, C.c_int, u8p, C.c_int]
    lib.ksw2_backtrack_left_end.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
    lib.ksw2_right_ext.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, ip]
    lib.ksw2_left_ext.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, ip]
    lib.edlib_align_HW.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, ip, ip, C.c_int]
    lib.get_bseq.argtypes = [C.c_char_p, C.c_int]
    lib.get_bseq.restype = C.POINTER(C.c_uint8)
    lib.get_rc_seq.argtypes = [C.c_char_p, C.c_int]
    lib.get_rc_seq.restype = C.c_void_p
    return lib


def default_para(**kw):
    p = Para()
    load().tho_para_init(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def u8(arr):
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def encode(seq: bytes):
    out = np.zeros(len(seq), dtype=np.uint8)
    load().tho_encode(seq, len(seq), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def ksw2_global_cigar(q, t):
    lib = load()
    qa, qp = u8(q); ta, tp = u8(t)
    n = C.c_int(); cg = C.POINTER(C.c_uint32)(); sc = C.c_int()
    iden = lib.tho_ksw2_global_cigar(qp, len(qa), tp, len(ta), C.byref(n), C.byref(cg), C.byref(sc))
    cigar = [cg[i] for i in range(n.value)]
    if n.value:
        lib.free(cg)
    return iden, cigar, sc.value


def ksw2_ext(q, t, left=False):
    lib = load()
    qa, qp = u8(q); ta, tp = u8(t)
    mq, mt = C.c_int(), C.c_int()
    (lib.tho_ksw2_left_ext if left else lib.tho_ksw2_right_ext)(qp, len(qa), tp, len(ta), C.byref(mq), C.byref(mt))
    return mq.value, mt.value


def edlib_hw(q: bytes, t: bytes, k: int):
    lib = load()
    s, e = C.c_int(-7), C.c_int(-7)
    ed = lib.tho_edlib_hw(q, len(q), t, len(t), C.byref(s), C.byref(e), k)
    return ed, s.value, e.value


def core(seq: bytes, para):
    """Run the oracle's per-read driver; returns a list of dict records."""
    lib = load()
    rr = ReadRes()
    lib.tho_core(seq, len(seq), C.byref(para), C.byref(rr))
    out = []
    for i in range(rr.n):
        c = rr.a[i]
        out.append(dict(seq=c.seq[:c.cons_len] if c.seq else b"", qual=(c.qual[:c.cons_len] if c.qual else None),
                        cons_len=c.cons_len, start=c.start, end=c.end, copy_num=c.copy_num, ave_match=c.ave_match,
                        full_length=c.full_length, sub_pos=[c.sub_pos[j] for j in range(c.pos_n)],
                        n_seqs=c.n_seqs, ext=(c.max_q_l, c.max_t_l, c.max_q_r, c.max_t_r)))
    lib.tho_read_res_free(C.byref(rr))
    return out


def stages(seq: bytes, para):
    """hits, chains (cells as (start,end)), par_pos per chain -- for stage-level parity tests."""
    lib = load()
    b = encode(seq)
    bp = b.ctypes.data_as(C.POINTER(C.c_uint8))
    hp = C.POINTER(C.c_uint64)()
    hn = lib.tho_collect_hits(bp, len(b), C.byref(para), C.byref(hp))
    hits = np.array([hp[i] for i in range(hn)], dtype=np.uint64)
    cr = ChainRes()
    lib.tho_chain(hp, hn, C.byref(para), C.byref(cr))
    chains = []
    # per-hit dp view (rows flattened in hit order): score, from (flattened hit index or -1)
    row_off = np.zeros(cr.tot_n + 1, dtype=np.int64)
    for i in range(cr.tot_n):
        row_off[i + 1] = row_off[i] + cr.row_size[i]
    score = np.zeros(hn, dtype=np.int32); frm = np.full(hn, -1, dtype=np.int32)
    for i in range(cr.tot_n):
        for j in range(cr.row_size[i]):
            d = cr.dp[i][j]
            score[row_off[i] + j] = d.score
            frm[row_off[i] + j] = -1 if d.from_i < 0 else row_off[d.from_i] + d.from_j
    for c in range(cr.ch_n):
        ch = cr.chain[c]
        cells = [int(row_off[ch.cell[x].i] + ch.cell[x].j) for x in range(ch.len)]
        n = C.c_int()
        pp = lib.tho_partition(bp, len(b), C.byref(cr), C.byref(ch), C.byref(para), C.byref(n))
        par = [pp[x] for x in range(n.value)]
        lib.free(pp)
        chains.append(dict(cells=cells, est_start=ch.est_start, est_period=ch.est_period, par_pos=par))
    lib.tho_chain_free(C.byref(cr))
    if hn:
        lib.free(hp)
    return dict(bseq=b, hits=hits, score=score, frm=frm, chains=chains)


POA_OPT_GAP_TIE, POA_OPT_BAND = 1, 2


def units_of(seq: bytes, sub_pos):
    """the units abpoa_gen_cons() slices out of a read for one run of boundaries (src/abpoa_cons.c:41-48), as code arrays"""
    b = encode(seq)
    L = len(seq)
    out = []
    for s, e in zip(sub_pos[:-1], sub_pos[1:]):
        if s < 0 or e < 0 or s >= L - 1 or e + 1 > L:
            continue
        out.append(b[s + 1:e + 1])
    return out


def _poa_call(fn, units, para, *extra):
    n = len(units)
    arrs = [np.ascontiguousarray(u, dtype=np.uint8) for u in units]
    ptrs = (C.POINTER(C.c_uint8) * n)(*[a.ctypes.data_as(C.POINTER(C.c_uint8)) for a in arrs])
    lens = (C.c_int * n)(*[len(a) for a in arrs])
    tot = sum(len(a) for a in arrs) + 1
    cons = np.zeros(tot, dtype=np.uint8); cov = np.zeros(tot, dtype=np.int32)
    cl = fn(n, ptrs, lens, C.byref(para), *extra, cons.ctypes.data_as(C.POINTER(C.c_uint8)), cov.ctypes.data_as(C.POINTER(C.c_int)))
    return bytes(b"ACGTN"[c] for c in cons[:cl]), cov[:cl].copy()


def poa_banded(units, para):
    """oracle/tho_poa.c (the restatement the HIP kernel is held to)"""
    return _poa_call(load().tho_poa_consensus, units, para)


def poa_exact(units, para, flags=0):
    """oracle/tho_poa_exact.c: full matrix, Kahn order; flags bit 0 = breadth-first column ranks, bit 1 = gap wins ties"""
    return _poa_call(load().tho_poa_exact, units, para, flags)
