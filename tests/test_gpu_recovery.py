"""The recovery paths ON THE GPU, against the oracle.

The reference has none of them: it allocates the worst case (src/tandem_chain.c:361-368) and abPOA never gives a job back
(src/abpoa_cons.c:86-87).  Here an arena, a table or a scratch estimate may be too small, or a POA job may have to be rerun on a slower
variant; every such path must end in the records the oracle computes.  The CPU emulation runs them too (tests/test_emul_parity.py), but it
cannot see what they get wrong on the device -- its scratch is generously padded and fresh memory happens to be zero there (round 3's
stale-table read passed under emulation) -- so each test forces one path with its environment knob on 256 reads of mixed shapes and runs
TWO batches on ONE handle: the second (the reads in reverse order) meets the first one's stale arenas, tables and counters."""
import numpy as np
import pytest

import oracle_lib as O
import th_util as U
import tidehunter_amd as TH

pytestmark = pytest.mark.gpu

# (period, copies, error): short / headline-like / long units (the relative-score POA class) / noisy
SHAPES = [(200, 20, 0.10), (1000, 6, 0.10), (90, 6, 0.08), (500, 8, 0.12), (60, 9, 0.10), (2600, 4, 0.10), (300, 5, 0.15), (35, 12, 0.12)]      # (2600: units long enough for the POA's score bounds)
N_READS = 256


@pytest.fixture(scope="module")
def batch():
    lib = TH.load_library()
    seqs = [TH.synth_read(lib, 77, i, *SHAPES[i % len(SHAPES)]) for i in range(N_READS)]
    para = O.default_para()
    want = U.oracle_records(seqs, para)
    assert sum(len(w) for w in want) >= N_READS * 3 // 4
    return seqs, para, want


def _run_twice(batch, label, stages=False):
    """two batches on one handle (the second in reverse order), both equal to the oracle; returns the handle's counters"""
    seqs, para, want = batch
    th = TH.TideHunter(device=0)
    try:
        got = th.process(seqs)
        if stages:
            U.stage_compare(th, seqs, para)
        assert U.assert_records_equal(seqs, got, para, label + " (first batch)", want=want) >= N_READS * 3 // 4
        rs, rw = seqs[::-1], want[::-1]
        got = th.process(rs)
        if stages:
            U.stage_compare(th, rs, para)
        assert U.assert_records_equal(rs, got, para, label + " (second batch, stale arenas)", want=rw) >= N_READS * 3 // 4
        st = th.stats()
    finally:
        th.close()
    assert st["batches"] == 2 and st["jobs_without_consensus"] == 0, st
    return st


def test_gpu_no_knob_baseline_takes_no_recovery_path(batch):
    st = _run_twice(batch, "no knob", stages=True)
    assert st["stage_reruns"] == 0, st


def test_gpu_chain_record_arena_grows(batch, monkeypatch):
    """TH_REC_CAP=1: one chain record of room -> TH_E_REC_OVERFLOW, extraction + partition rerun with a larger arena, in both batches"""
    monkeypatch.setenv("TH_REC_CAP", "1")
    st = _run_twice(batch, "TH_REC_CAP=1", stages=True)
    assert st["stage_reruns"] >= 2, st


def test_gpu_job_tables_grow(batch, monkeypatch):
    """TH_JOB_CAP=1: job / unit / par_pos tables of one entry -> TH_E_JOB_CAP, nothing written, the build reruns with tables sized from the totals"""
    monkeypatch.setenv("TH_JOB_CAP", "1")
    st = _run_twice(batch, "TH_JOB_CAP=1", stages=True)
    assert st["stage_reruns"] >= 2, st


def test_gpu_partition_regions_grow(batch, monkeypatch):
    """TH_PAR_MUL=1: a first pass with four-entry par_pos regions -> TH_E_PART_OVERFLOW, the partition reruns with larger regions"""
    monkeypatch.setenv("TH_PAR_MUL", "1")
    st = _run_twice(batch, "TH_PAR_MUL=1", stages=True)
    assert st["stage_reruns"] >= 2, st


def test_gpu_all_arenas_too_small_together(batch, monkeypatch):
    monkeypatch.setenv("TH_REC_CAP", "1"); monkeypatch.setenv("TH_PAR_MUL", "1"); monkeypatch.setenv("TH_JOB_CAP", "1")
    st = _run_twice(batch, "TH_REC_CAP=1 TH_PAR_MUL=1 TH_JOB_CAP=1", stages=True)
    assert st["stage_reruns"] >= 4, st


def test_gpu_poa_band_failure_is_rerun_without_a_band(batch, monkeypatch):
    """TH_POA_TEST_FAIL=2: every second job reports "band lost the path" on the banded pass and is rerun with full rows on the 32-bit variant
    (the band never matters at these error rates, so the records are the oracle's)"""
    monkeypatch.setenv("TH_POA_TEST_FAIL", "2")
    st = _run_twice(batch, "TH_POA_TEST_FAIL=2")
    assert st["poa_jobs_rerun"] >= st["jobs"] // 4, st


def test_gpu_poa_node_capacity_overflow_is_rerun(batch, monkeypatch):
    """TH_POA_TEST_SMALLCAP=1: node arrays carved just above the longest unit, so every graph outgrows them at its first merge (status 1) and is
    rerun with twice the factor"""
    monkeypatch.setenv("TH_POA_TEST_SMALLCAP", "1")
    st = _run_twice(batch, "TH_POA_TEST_SMALLCAP=1")
    assert st["poa_jobs_rerun"] >= st["jobs"] // 2, st


def test_gpu_poa_forced_long_variant(batch, monkeypatch):
    """TH_POA_FORCE_LONG=1: every job on thpoa_l (relative int16 scores, three-trip ring rows), short units included"""
    monkeypatch.setenv("TH_POA_FORCE_LONG", "1")
    _run_twice(batch, "TH_POA_FORCE_LONG=1")


def test_gpu_poa_reruns_and_arena_growth_together(batch, monkeypatch):
    monkeypatch.setenv("TH_REC_CAP", "1"); monkeypatch.setenv("TH_POA_TEST_SMALLCAP", "1"); monkeypatch.setenv("TH_POA_TEST_FAIL", "3")
    st = _run_twice(batch, "TH_REC_CAP=1 TH_POA_TEST_SMALLCAP=1 TH_POA_TEST_FAIL=3", stages=True)
    assert st["stage_reruns"] >= 2 and st["poa_jobs_rerun"] >= st["jobs"] // 2, st
