"""Reference-generated vectors (tests/golden/ref_ksw2_vectors.npz, ref_edlib_vectors.npz: written by tests/golden/make_ref_vectors.py
from the reference's OWN ksw2 / edlib code, oracle/_ref/libthref.so) against

  * the oracle (CPU, any box -- the files travel, /root/reference does not),
  * the kernel sources under the CPU emulation, through the raw-pair hooks (a subset: the emulation is slow),
  * the HIP kernels on the GPU (`-m gpu`), every engine the alignment kernel has: strip engine, pruned second band, anti-diagonal band
    engine at one to six registers per lane, the deferred strip launch, flank extensions on the anti-diagonal engine / pruned strips /
    full matrix, the partition's full-matrix engine with the CIGAR functionals, and both adapter searches.

src/ksw2_align.c:88-173 (ksw2_backtrack_left_end, ksw2_global, ksw2_global_with_cigar, ksw2_right_ext, ksw2_left_ext),
src/edlib_align.c:73-85 (edlib_align_HW)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import th_util as U

GOLD = os.path.join(U.ROOT, "tests", "golden")
CLS = {"short": 0, "strip": 1, "kb": 2, "long": 3}


@pytest.fixture(scope="module")
def kv():
    with np.load(os.path.join(GOLD, "ref_ksw2_vectors.npz")) as z:
        return {k: z[k] for k in z.files}          # (an NpzFile inflates an array at EVERY access)


@pytest.fixture(scope="module")
def ev():
    with np.load(os.path.join(GOLD, "ref_edlib_vectors.npz")) as z:
        return {k: z[k] for k in z.files}


def _pair(v, i):
    return v["q_cat"][v["q_off"][i]:v["q_off"][i + 1]], v["t_cat"][v["t_off"][i]:v["t_off"][i + 1]]


# ------------------------------------------------------------------------------------------------ the oracle
def test_oracle_ksw2_matches_reference_vectors(kv):
    lib = O.load()
    n = len(kv["iden"])
    assert n > 2500
    for i in range(n):
        q, t = _pair(kv, i)
        if kv["cls"][i] == CLS["long"] and i % 4:
            continue                                  # (a quarter of the 5000-base pairs: the scalar oracle keeps a traceback matrix)
        iden, cigar, _ = O.ksw2_global_cigar(q, t)
        assert iden == kv["iden"][i], i
        arr = (C.c_uint32 * len(cigar))(*cigar)
        for x, want in zip(kv["q_left_ext"][i], kv["t_back"][i]):
            assert lib.tho_ksw2_backtrack_left_end(len(cigar), arr, len(q), len(t), int(x)) == want, (i, x)
        assert O.ksw2_ext(q, t, True) == tuple(kv["left_ext"][i]), i
        assert O.ksw2_ext(q, t, False) == tuple(kv["right_ext"][i]), i


def _queries(ev):
    qo = ev["q_off"]
    return [bytes(ev["queries"][qo[j]:qo[j + 1]]) for j in range(len(qo) - 1)]


def test_oracle_edlib_matches_reference_vectors(ev):
    qs, ks = _queries(ev), ev["k"]
    to, co = ev["t_off"], ev["c_off"]
    n_hit = 0
    for i in range(len(to) - 1):
        t = bytes(ev["t_cat"][to[i]:to[i + 1]])
        for j, q in enumerate(qs):
            got = O.edlib_hw(q, t, int(ks[j]))
            want = tuple(ev["res"][i, j])
            assert (got if got[0] != -1 else (-1, -1, -1)) == want, (i, j)
            n_hit += want[0] >= 0
    for i in range(len(co) - 1):
        t = bytes(b"ACGTN"[c] for c in ev["c_cat"][co[i]:co[i + 1]]) * 2
        for j, q in enumerate(qs):
            got = O.edlib_hw(q, t, int(ks[j]))
            assert (got if got[0] != -1 else (-1, -1, -1)) == tuple(ev["res_doubled"][i, j]), (i, j)
    assert n_hit > 1000


# ------------------------------------------------------------------------------------------------ the device engines, through the hooks
def _bind(lib):
    u8p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    lib.th_debug_align_pairs.argtypes = [C.c_int, C.c_int, C.c_int, u8p, ip, u8p, ip, ip, ip, ip]
    lib.th_debug_global_pairs.argtypes = [C.c_int, C.c_int, u8p, ip, u8p, ip, ip, C.c_int, ip, ip, C.c_int]
    lib.th_debug_adapter.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), ip, ip, C.c_int, u8p, ip, ip]
    return lib


def _subset(v, idx):
    """the pairs idx of a vector file, concatenated again"""
    qs = [v["q_cat"][v["q_off"][i]:v["q_off"][i + 1]] for i in idx]
    ts = [v["t_cat"][v["t_off"][i]:v["t_off"][i + 1]] for i in idx]
    qc = np.ascontiguousarray(np.concatenate(qs), dtype=np.uint8); tc = np.ascontiguousarray(np.concatenate(ts), dtype=np.uint8)
    qo = np.concatenate([[0], np.cumsum([len(a) for a in qs])]).astype(np.int32); to = np.concatenate([[0], np.cumsum([len(a) for a in ts])]).astype(np.int32)
    return qc, qo, tc, to


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def align_pairs(lib, v, idx, kind, cfg):
    """cfg: dict(variant, mode, wdiv, wadd, wtol, xpct, defer) -> (results, stats)"""
    qc, qo, tc, to = _subset(v, idx)
    n = len(idx)
    c = np.array([cfg.get("variant", 0), cfg.get("mode", 1), cfg.get("wdiv", 10), cfg.get("wadd", 24), cfg.get("wtol", 16), cfg.get("xpct", 36), cfg.get("defer", 0)], dtype=np.int32)
    out = np.full(n if kind == 0 else 2 * n, -12345, dtype=np.int32); st = np.zeros(8, dtype=np.int32)
    rc = lib.th_debug_align_pairs(0, kind, n, _p(qc, C.c_uint8), _p(qo, C.c_int), _p(tc, C.c_uint8), _p(to, C.c_int), _p(c, C.c_int), _p(out, C.c_int), _p(st, C.c_int))
    assert rc == 0
    return (out if kind == 0 else out.reshape(n, 2)), st


def global_pairs(lib, v, idx, banded=0):
    qc, qo, tc, to = _subset(v, idx)
    n = len(idx)
    x = np.ascontiguousarray(v["q_left_ext"][idx], dtype=np.int32)
    iden = np.full(n, -12345, dtype=np.int32); tb = np.full((n, x.shape[1]), -12345, dtype=np.int32)
    rc = lib.th_debug_global_pairs(0, n, _p(qc, C.c_uint8), _p(qo, C.c_int), _p(tc, C.c_uint8), _p(to, C.c_int), _p(x, C.c_int), x.shape[1], _p(iden, C.c_int), _p(tb, C.c_int), banded)
    assert rc == 0
    return iden, tb


def adapter(lib, ev, group, doubled, rows):
    """queries 4 * group .. 4 * group + 3 against the targets `rows` (byte targets, or consensus codes searched over the doubled text)"""
    qs, ks = _queries(ev), ev["k"]
    q4 = qs[4 * group:4 * group + 4]
    arr = (C.c_char_p * 4)(*q4); ql = np.array([len(q) for q in q4], dtype=np.int32); kk = np.ascontiguousarray(ks[4 * group:4 * group + 4], dtype=np.int32)
    cat, off = (ev["c_cat"], ev["c_off"]) if doubled else (ev["t_cat"], ev["t_off"])
    ts = [cat[off[i]:off[i + 1]] for i in rows]
    tc = np.ascontiguousarray(np.concatenate(ts), dtype=np.uint8); to = np.concatenate([[0], np.cumsum([len(t) for t in ts])]).astype(np.int32)
    out = np.full((len(rows), 4, 3), -12345, dtype=np.int32)
    rc = lib.th_debug_adapter(0, 1 if doubled else 0, arr, _p(ql, C.c_int), _p(kk, C.c_int), len(rows), _p(tc, C.c_uint8), _p(to, C.c_int), _p(out, C.c_int))
    assert rc == 0
    return out


def _idx(v, names, step=1, maxlen=None):
    sel = [i for i in range(len(v["iden"])) if v["cls"][i] in [CLS[c] for c in names]]
    if maxlen:
        sel = [i for i in sel if v["q_off"][i + 1] - v["q_off"][i] <= maxlen and v["t_off"][i + 1] - v["t_off"][i] <= maxlen]
    return np.array(sel[::step], dtype=np.int64)


@pytest.fixture(scope="module")
def emu_lib():
    return _bind(C.CDLL(U.emul_lib_path()))


def test_emul_engines_match_reference_vectors(emu_lib, kv, ev):
    """the kernel sources under the CPU emulation (a thin sample: it runs every wave as 64 fibers)"""
    idx = np.concatenate([_idx(kv, ["short"], step=40, maxlen=120), _idx(kv, ["strip"], step=30, maxlen=230)])
    for cfg in (dict(mode=0, wdiv=100000, wadd=2), dict(mode=1), dict(mode=1, variant=1, defer=1, wdiv=100000, wadd=3)):
        got, st = align_pairs(emu_lib, kv, idx, 0, cfg)
        assert st[0] == 0 and (got == kv["iden"][idx]).all(), (cfg, np.nonzero(got != kv["iden"][idx]))
    for kind, key in ((1, "left_ext"), (2, "right_ext")):
        got, st = align_pairs(emu_lib, kv, idx[::2], kind, dict(mode=1))
        assert st[0] == 0 and (got == kv[key][idx[::2]]).all(), kind
    for banded in (0, 1):
        iden, tb = global_pairs(emu_lib, kv, idx[::3], banded)
        assert (iden == kv["iden"][idx[::3]]).all() and (tb == kv["t_back"][idx[::3]]).all(), banded
    rows = list(range(0, 40, 3))
    for g in (0, 2):
        assert (adapter(emu_lib, ev, g, False, rows) == ev["res"][rows][:, 4 * g:4 * g + 4]).all()
        assert (adapter(emu_lib, ev, g, True, rows) == ev["res_doubled"][rows][:, 4 * g:4 * g + 4]).all()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu_lib():
    import tidehunter_amd as TH
    return _bind(TH.load_library())


# unit-vs-consensus alignments (ksw2_global): which engine computes a pair never changes its result
ALIGN_CFGS = [
    ("strip engine only, default first band", dict(mode=0)),
    ("strip engine only, first band of 2: nearly every pair goes through the pruned second band (slb)", dict(mode=0, wdiv=100000, wadd=2)),
    ("strip engine only, first band of 6", dict(mode=0, wdiv=100000, wadd=6)),
    ("anti-diagonal engine first, defaults, small-area variant (<= 2 registers)", dict(mode=1)),
    ("anti-diagonal engine, verify every banded result on the device (TH_ALN_MODE=2)", dict(mode=2)),
    ("anti-diagonal engine, one register, narrow request", dict(mode=1, wdiv=100000, wadd=20, wtol=0)),
    ("anti-diagonal engine, two registers", dict(mode=1, wdiv=100000, wadd=190, wtol=0)),
    ("big-area variant, three registers", dict(mode=1, variant=2, wdiv=100000, wadd=320, wtol=0)),
    ("big-area variant, four registers", dict(mode=1, variant=2, wdiv=100000, wadd=440, wtol=0)),
    ("big-area variant, five registers", dict(mode=1, variant=2, wdiv=100000, wadd=570, wtol=0)),
    ("big-area variant, six registers", dict(mode=1, variant=2, wdiv=100000, wadd=700, wtol=0)),
    ("16 KB variant, defaults, deferred strip launch", dict(mode=1, variant=1, defer=1)),
    ("24 KB variant, tiny first band, everything uncertified is deferred to the strip launch", dict(mode=1, variant=2, defer=1, wdiv=100000, wadd=3, wtol=0)),
    ("24 KB variant without the deferred launch: the strip engines run in place", dict(mode=1, variant=2, defer=0, wdiv=100000, wadd=3, wtol=0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("label,cfg", ALIGN_CFGS, ids=[c[0].split(",")[0].replace(" ", "_") + "_%d" % i for i, c in enumerate(ALIGN_CFGS)])
def test_gpu_unit_alignments_match_reference_vectors(gpu_lib, kv, label, cfg):
    idx = np.arange(len(kv["iden"]))
    got, st = align_pairs(gpu_lib, kv, idx, 0, cfg)
    bad = np.nonzero(got != kv["iden"])[0]
    assert st[0] == 0, "%s: device error flags %d" % (label, st[0])
    assert len(bad) == 0, "%s: %d pairs differ from the reference, first %s" % (label, len(bad), bad[:5])
    if cfg.get("defer"):
        assert st[1] > 0, "%s: nothing was deferred" % label


EXT_CFGS = [
    ("anti-diagonal engine / pruned strips / full matrix as the pipeline picks them", dict(mode=1)),
    ("every engine verified against the full matrix on the device", dict(mode=2)),
    ("full matrix only", dict(mode=0)),
    ("expected score of 90 % of the flank: narrow first bands, retries through wider ones and the pruned strip engine", dict(mode=1, xpct=10)),
    ("expected score of 5 %: the widest band first", dict(mode=1, xpct=95)),
    ("big-area variant", dict(mode=1, variant=2)),
    ("four-register probe for the long flanks (round 3's; the default is one: upper bits of xpct)", dict(mode=1, xpct=36 | 4 << 8)),
    ("two-register probe, verified on the device", dict(mode=2, xpct=36 | 2 << 8)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("label,cfg", EXT_CFGS, ids=["ext_%d" % i for i in range(len(EXT_CFGS))])
def test_gpu_flank_extensions_match_reference_vectors(gpu_lib, kv, label, cfg):
    idx = np.arange(len(kv["iden"]))
    for kind, key in ((1, "left_ext"), (2, "right_ext")):
        got, st = align_pairs(gpu_lib, kv, idx, kind, cfg)
        bad = np.nonzero((got != kv[key]).any(axis=1))[0]
        assert st[0] == 0, "%s: device error flags %d" % (label, st[0])
        assert len(bad) == 0, "%s (%s): %d pairs differ from the reference, first %s: got %s want %s" % (label, key, len(bad), bad[:5], got[bad[:3]], kv[key][bad[:3]])


def _noisy(rng, codes, err):
    """substitutions, insertions and deletions in equal parts, as th_synth_read draws them"""
    out = []
    for b in codes:
        if rng.random() < err:
            k = int(rng.integers(0, 3))
            if k == 0:
                out.append((int(b) + 1 + int(rng.integers(0, 3))) & 3)
            elif k == 1:
                out.append(int(b)); out.append(int(rng.integers(0, 4)))
        else:
            out.append(int(b))
    return np.array(out, dtype=np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(mode=1, variant=3), dict(mode=2, variant=3), dict(mode=1, variant=3, xpct=36 | 1 << 12), dict(mode=1), dict(mode=1, variant=1), dict(mode=1, variant=3, xpct=36 | 2 << 8)],
                         ids=["longest_first_instantiation", "verified_on_device", "long_flank_path_switched_off", "plain_instantiation", "16k_variant", "two_register_probe"])
def test_gpu_flank_longer_than_the_staging_area_is_probed_on_its_prefix(gpu_lib, cfg):
    """round 5: a flank that does not fit th_k_align's staging area next to the consensus (config 4: a 5000-base consensus and 7000 ... 10 000 flank bases when the first unit
    boundary was found a period late) went to the full matrix unprobed; it is probed on the prefix that fits -- a lower bound of the whole maximum -- and finished on the pruned
    strips.  Results against the oracle's ksw2_left_ext / ksw2_right_ext (src/ksw2_align.c:175-215) and against the device's own full matrix (mode 0): flanks that continue the
    repeat past a whole period, one that stops matching half way, one of random bases"""
    rng = np.random.default_rng(11)
    pairs = {1: [], 2: []}
    for cl, extra, err in ((5000, 2600, 0.15), (4800, 4300, 0.12), (5200, 1900, 0.15), (3000, 7200, 0.10)):
        cons = rng.integers(0, 4, cl).astype(np.uint8)
        rep = np.concatenate([cons] * 4)
        pairs[2].append((cons, _noisy(rng, rep[:cl + extra], err)))                      # right: the flank starts where the consensus starts
        pairs[1].append((cons, _noisy(rng, rep[len(rep) - cl - extra:], err)))           # left: the flank ends where the consensus ends
    cons = rng.integers(0, 4, 5000).astype(np.uint8)
    half = np.concatenate([_noisy(rng, cons[:2500], 0.1), rng.integers(0, 4, 5500).astype(np.uint8)])
    pairs[2].append((cons, half)); pairs[1].append((cons, half[::-1].copy()))
    pairs[2].append((cons, rng.integers(0, 4, 7600).astype(np.uint8))); pairs[1].append((cons, rng.integers(0, 4, 7600).astype(np.uint8)))
    for kind in (1, 2):
        v = {"q_cat": np.concatenate([q for q, _ in pairs[kind]]), "t_cat": np.concatenate([t for _, t in pairs[kind]]),
             "q_off": np.concatenate([[0], np.cumsum([len(q) for q, _ in pairs[kind]])]).astype(np.int32), "t_off": np.concatenate([[0], np.cumsum([len(t) for _, t in pairs[kind]])]).astype(np.int32)}
        idx = np.arange(len(pairs[kind]))
        got, st = align_pairs(gpu_lib, v, idx, kind, cfg)
        assert st[0] == 0, "device error flags %d" % st[0]
        # (the prefix probe lives in the longest-first instantiation only -- variant 3 of the hook, which hands the kernel its pass queues; st[4] counts the
        # flanks that took th_aln_long_flank: all six must when the path is on, none when bit 12 of xpct switches it off or another instantiation runs)
        took = int(st[4])
        if cfg.get("variant") == 3 and not (cfg.get("xpct", 0) >> 12) & 1:
            assert took == len(idx), "th_aln_long_flank ran for %d of %d flanks" % (took, len(idx))
        else:
            assert took == 0
        full, st0 = align_pairs(gpu_lib, v, idx, kind, dict(mode=0))
        assert st0[0] == 0 and (got == full).all(), (kind, got, full)
        for i in (0, 3, 4):
            q, t = pairs[kind][i]
            assert tuple(int(x) for x in got[i]) == O.ksw2_ext(q, t, left=(kind == 1)), (kind, i)


@pytest.mark.gpu
@pytest.mark.parametrize("banded", [0, 1], ids=["full_matrix", "certified_band_then_pruned_band"])
def test_gpu_global_with_cigar_backprojection_matches_reference_vectors(gpu_lib, kv, banded):
    """the partition's engines (th_ksw_global_wave; th_ksw_global_r_wave: the packed strip engine with the back-projection functional): iden_n and
    ksw2_backtrack_left_end for four q_left_ext per pair"""
    idx = np.arange(len(kv["iden"]))
    iden, tb = global_pairs(gpu_lib, kv, idx, banded)
    assert (iden == kv["iden"]).all(), np.nonzero(iden != kv["iden"])[0][:5]
    bad = np.nonzero((tb != kv["t_back"]).any(axis=1))[0]
    assert len(bad) == 0, (bad[:5], tb[bad[:3]], kv["t_back"][bad[:3]], kv["q_left_ext"][bad[:3]])


@pytest.mark.gpu
def test_gpu_adapter_searches_match_reference_vectors(gpu_lib, ev):
    n_t, n_c = len(ev["t_off"]) - 1, len(ev["c_off"]) - 1
    for g in range(len(ev["k"]) // 4):
        got = adapter(gpu_lib, ev, g, False, list(range(n_t)))
        want = ev["res"][:, 4 * g:4 * g + 4]
        assert (got == want).all(), ("th_k_adapter_reads", g, np.argwhere((got != want).any(axis=2))[:5])
        got = adapter(gpu_lib, ev, g, True, list(range(n_c)))
        want = ev["res_doubled"][:, 4 * g:4 * g + 4]
        assert (got == want).all(), ("th_k_adapter", g, np.argwhere((got != want).any(axis=2))[:5])
