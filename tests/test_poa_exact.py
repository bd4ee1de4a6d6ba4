"""Stage 4 (POA) without libabpoa: what CAN be checked on the CPU.

abPOA is an empty submodule of the reference (.gitmodules:1-4), so oracle/tho_poa.c -- the restatement the HIP kernel
is held to -- has no reference output to be pinned against.  These tests tie it down from the other side instead:

  * oracle/tho_poa_exact.c is a second, independent POA (full matrix, Kahn order recomputed per sequence, different data
    structures).  On samples of the shapes of BASELINE.json's configs the adaptive band and the incremental node order of
    tho_poa.c do not change a consensus: banded == exact, with consensus columns ranked depth-first or breadth-first.
  * At error rates far above the configs' (20-30 %) the adaptive band -- which is part of the published algorithm --
    does lose the optimum now and then; the ORDER still never matters (tho_poa.c unbanded == exact).
  * README.md:232 is an ANCHOR, NOT A PIN: the consensus it prints (203 bases) belongs to an older partition of the same
    read (its subPos column is stale), so only a bounded edit distance is asserted.
The larger sample (>= 2000 jobs per config, against the HIP kernel's output) runs on the GPU box: tests/test_gpu_scale.py.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as O
import th_util as U

README_232_CONS = ("ACTAATAAGATCAACAGAATCAGAGTAGATAGTTCCTTGATCGGAACCAAAGGACCCCGTGCCTCAATCTCTATCCTGATGTCATGGGAGTCCTAGCAAAGCTATAGACTCAAGCAAGG"
                   "CTTGGGGTCCTTTATGGAACCCAAGGATGACTCAGCAATAAAATATTTTGGTTTTGGTTTATAAAAAAAAAAAAAAAAAAAAAA")


def edit_distance(a: bytes, b: bytes) -> int:
    prev = np.arange(len(b) + 1)
    bb = np.frombuffer(b, dtype=np.uint8)
    for i, ca in enumerate(a, 1):
        cur = np.empty_like(prev)
        cur[0] = i
        sub = prev[:-1] + (bb != ca)
        best = np.minimum(sub, prev[1:] + 1)
        for j in range(1, len(b) + 1):           # insertion chain
            cur[j] = min(best[j - 1], cur[j - 1] + 1)
        prev = cur
    return int(prev[-1])


def _jobs(period, copies, err, n, seed):
    """(units, banded consensus record) of every POA job (> 2 units) the oracle runs on n synthetic reads"""
    lib = U.synth_lib()
    para = O.default_para()
    out = []
    for i in range(n):
        s = U.synth_read(lib, seed, i, period, copies, err)
        for r in O.core(s, para):
            u = O.units_of(s, r["sub_pos"])
            if len(u) > 2:
                out.append((u, r["seq"]))
    return out, para


@pytest.mark.parametrize("period,copies,err,n,seed", [
    (200, 20, 0.10, 48, 2),       # config 2 shape
    (1000, 10, 0.10, 16, 1),      # config 3 shape (headline)
    (5000, 4, 0.15, 2, 3),        # config 4 shape
    (90, 12, 0.12, 40, 17),
])
def test_band_and_order_never_change_the_consensus_on_config_shapes(period, copies, err, n, seed):
    jobs, para = _jobs(period, copies, err, n, seed)
    assert len(jobs) >= n * 3 // 4

    def one(job):
        u, rec = job
        b, bcov = O.poa_banded(u, para)
        d, dcov = O.poa_exact(u, para, 0)
        q, qcov = O.poa_exact(u, para, 1)
        return rec == b and b == d and d == q and list(bcov) == list(dcov) == list(qcov)

    with ThreadPoolExecutor(8) as ex:
        ok = list(ex.map(one, jobs))
    # (on these samples the band is harmless too; at scale it is not always: tests/test_gpu_scale.py found 1 of 2053 config-2 jobs)
    assert all(ok), "%d of %d jobs differ between tho_poa.c and the exact POA" % (ok.count(False), len(ok))


def test_order_never_matters_even_where_the_band_does():
    """20-30 % error: the adaptive band may miss the optimum (it is part of the algorithm); the node order may not"""
    lib = O.load()
    n_jobs = band_diff = 0
    try:
        for period, copies, err, n, seed in ((500, 8, 0.20, 40, 9), (300, 8, 0.30, 120, 8), (700, 6, 0.25, 60, 7)):
            jobs, para = _jobs(period, copies, err, n, seed)
            for u, rec in jobs:
                lib.tho_poa_set_opt(O.POA_OPT_BAND, 0)
                ub, _ = O.poa_banded(u, para)
                lib.tho_poa_set_opt(O.POA_OPT_BAND, 1)
                d, _ = O.poa_exact(u, para, 0)
                q, _ = O.poa_exact(u, para, 1)
                assert ub == d == q, "node order changed a consensus"
                band_diff += rec != ub
                n_jobs += 1
    finally:
        lib.tho_poa_set_opt(O.POA_OPT_BAND, 1)
    assert n_jobs >= 40
    assert band_diff <= n_jobs // 4          # the band is usually harmless even here


def _full_length_consensus():
    seq = U.read_fastx(os.path.join(U.DATA, "full_length.fa"))[0][1]
    para = O.default_para()
    O.load().tho_para_load_adapters(para, os.path.join(U.DATA, "5prime.fa").encode(), os.path.join(U.DATA, "3prime.fa").encode())
    (rec,) = O.core(seq, para)
    return rec


def test_readme_232_anchor_not_pin():
    """anchor, not pin: README.md:232 shows a 203-base consensus for an OLDER partition of test_data/full_length.fa"""
    rec = _full_length_consensus()
    assert rec["full_length"] == 1 and "%.1f" % rec["copy_num"] == "8.8"
    d = edit_distance(rec["seq"], README_232_CONS.encode())
    assert abs(rec["cons_len"] - 203) <= 4 and d <= 6, (rec["cons_len"], d)
    assert abs(rec["ave_match"] - 95.7) < 1.0


def test_gap_tie_rule_is_a_named_switch():
    """THO_POA_GAP_WINS_TIE (tho.h) / TH_POA_GAP_WINS_TIE (th_k_poa.hip): one constant each to pin when abPOA's source is available.
    The default (a base as frequent as the gap stays) is the variant closer to the README:232 anchor."""
    lib = O.load()
    base_wins = _full_length_consensus()
    try:
        lib.tho_poa_set_opt(O.POA_OPT_GAP_TIE, 1)
        gap_wins = _full_length_consensus()
    finally:
        lib.tho_poa_set_opt(O.POA_OPT_GAP_TIE, 0)
    d0 = edit_distance(base_wins["seq"], README_232_CONS.encode())
    d1 = edit_distance(gap_wins["seq"], README_232_CONS.encode())
    assert gap_wins["cons_len"] <= base_wins["cons_len"] and d0 <= d1
    # both implementations honour the switch the same way
    seq = U.read_fastx(os.path.join(U.DATA, "full_length.fa"))[0][1]
    para = O.default_para()
    u = O.units_of(seq, base_wins["sub_pos"])
    try:
        lib.tho_poa_set_opt(O.POA_OPT_GAP_TIE, 1)
        a, _ = O.poa_banded(u, para)
    finally:
        lib.tho_poa_set_opt(O.POA_OPT_GAP_TIE, 0)
    b, _ = O.poa_exact(u, para, 2)
    assert a == b
    src = open(os.path.join(U.ROOT, "tidehunter_amd", "csrc", "th_k_poa.hip")).read()
    assert "#define TH_POA_GAP_WINS_TIE 0" in src
