"""tidehunter_amd -- Python binding (ctypes) of the MI355X-native TideHunter hot path.

The compute lives in tidehunter_amd/libtidehunter_amd.so (hand-written HIP for gfx950 behind the
C-ABI of include/tidehunter_amd.h).  This module is plumbing: it loads the library, mirrors the
parameter/result structs and exposes the batch seam.  There is no CPU fallback: if the HIP library
is missing or no GPU is present, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtidehunter_amd.so")


class Para(C.Structure):
    """th_para_t == mini_tandem_para (src/tidehunter.h:47-61)"""
    _fields_ = [("k", C.c_int64), ("w", C.c_int64), ("hpc", C.c_int), ("min_copy", C.c_int), ("min_cov", C.c_int),
                ("max_div", C.c_double), ("div_exp", C.c_double), ("min_frac", C.c_double),
                ("min_p", C.c_int64), ("max_p", C.c_int64), ("max_range", C.c_int64),
                ("match", C.c_int), ("mismatch", C.c_int), ("gap_open1", C.c_int), ("gap_open2", C.c_int),
                ("gap_ext1", C.c_int), ("gap_ext2", C.c_int), ("ada_match_rat", C.c_float),
                ("five_fn", C.c_void_p), ("five_seq", C.c_void_p), ("five_rc_seq", C.c_void_p), ("five_len", C.c_int),
                ("three_fn", C.c_void_p), ("three_seq", C.c_void_p), ("three_rc_seq", C.c_void_p), ("three_len", C.c_int),
                ("out_fmt", C.c_int), ("min_len", C.c_int), ("only_unit", C.c_int), ("only_longest", C.c_int),
                ("only_full_length", C.c_int), ("single_copy", C.c_int), ("cons_out", C.c_void_p), ("detail_fp", C.c_void_p),
                ("n_thread", C.c_int)]


class KString(C.Structure):
    """th_kstring_t == kstring_t (src/kseq.h:84-87)"""
    _fields_ = [("l", C.c_size_t), ("m", C.c_size_t), ("s", C.POINTER(C.c_char))]


class Seq(C.Structure):
    """th_seq_t == seq_t (src/seq.h:6-8)"""
    _fields_ = [("name", KString), ("seq", KString), ("comment", KString), ("qual", KString)]


class TandemSeq(C.Structure):
    """th_tandem_seq_t == tandem_seq_t (src/tidehunter.h:63-72), identical layout"""
    _fields_ = [("cons_seq", C.POINTER(Seq)),
                ("cons_n", C.c_int), ("cons_m", C.c_int),
                ("cons_start", C.POINTER(C.c_int)), ("cons_end", C.POINTER(C.c_int)), ("cons_len", C.POINTER(C.c_int)),
                ("copy_num", C.POINTER(C.c_double)), ("ave_match", C.POINTER(C.c_double)),
                ("full_length", C.POINTER(C.c_int8)),
                ("sub_pos", C.POINTER(C.POINTER(C.c_int))), ("pos_n", C.POINTER(C.c_int)), ("pos_m", C.POINTER(C.c_int)),
                ("cons_score", C.POINTER(C.c_int))]


def load_library(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError("tidehunter_amd: %s is missing -- build it with __graft_entry__.build() "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.th_init_para.restype = C.POINTER(Para)
    lib.th_free_para.argtypes = [C.POINTER(Para)]
    lib.th_load_adapters.argtypes = [C.POINTER(Para)]
    lib.th_alloc_tandem_seq.argtypes = [C.c_int]
    lib.th_alloc_tandem_seq.restype = C.POINTER(TandemSeq)
    lib.th_free_tandem_seq.argtypes = [C.POINTER(TandemSeq), C.c_int]
    lib.th_init.argtypes = [C.POINTER(Para), C.c_int]
    lib.th_init.restype = C.c_void_p
    lib.th_destroy.argtypes = [C.c_void_p]
    pp, ip = C.POINTER(C.c_char_p), C.POINTER(C.c_int)
    lib.th_process_batch.argtypes = [C.c_void_p, pp, ip, C.c_int, C.POINTER(TandemSeq)]
    lib.th_batch_upload.argtypes = [C.c_void_p, pp, ip, C.c_int]
    lib.th_batch_run.argtypes = [C.c_void_p]
    lib.th_batch_collect.argtypes = [C.c_void_p, C.POINTER(TandemSeq)]
    lib.th_batch_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lib.th_batch_launches.argtypes = [C.c_void_p]
    lib.th_batch_syncs.argtypes = [C.c_void_p]
    lib.th_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
    lib.th_batch_run_steps.argtypes = [C.c_void_p, C.c_int, C.POINTER(TandemSeq), C.POINTER(C.c_double)]
    lib.th_process_batches.argtypes = [C.c_void_p, C.c_int, C.POINTER(pp), C.POINTER(ip), ip, C.POINTER(C.POINTER(TandemSeq))]
    lib.th_run_file.argtypes = [C.c_char_p, C.POINTER(Para), C.c_void_p, C.c_int, C.c_int]
    lib.th_synth_read.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_char_p, C.c_int]
    lib.th_synth_adapter.argtypes = [C.c_uint64, C.c_int, C.c_char_p, C.c_int]
    lib.th_synth_read_c5.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.c_int]
    lib.th_parse_file.argtypes = [C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(C.c_ulonglong)]
    lib.th_parse_file.restype = C.c_longlong
    lib.th_ticket_open.argtypes = [C.c_char_p, C.c_int]
    lib.th_ticket_open.restype = C.c_void_p
    lib.th_ticket_next.argtypes = [C.c_void_p, C.c_int]
    lib.th_ticket_next.restype = C.c_longlong
    lib.th_ticket_close.argtypes = [C.c_void_p, C.c_int]
    lib.th_last_error.restype = C.c_char_p
    lib.th_version.restype = C.c_char_p
    lib.th_debug_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
    lib.th_debug_stage.restype = C.c_longlong
    return lib


class TideHunter:
    """One handle = one GPU.  Mirrors the reference's per-chunk call surface (src/main.c:402-425)."""

    def __init__(self, device=0, lib_path=None, adapters=None, **para):
        self.lib = load_library(lib_path)
        self._c = C.CDLL(None)
        self._c.strdup.restype = C.c_void_p
        self._c.strdup.argtypes = [C.c_char_p]
        self.para = self.lib.th_init_para()
        for k, v in para.items():
            setattr(self.para.contents, k, v)
        if adapters is not None:
            self.para.contents.five_fn = self._c.strdup(os.fsencode(adapters[0]))
            self.para.contents.three_fn = self._c.strdup(os.fsencode(adapters[1]))
            if self.lib.th_load_adapters(self.para) != 0:
                raise RuntimeError(self.lib.th_last_error().decode())
        self.h = self.lib.th_init(self.para, device)
        if not self.h:
            raise RuntimeError("tidehunter_amd: " + self.lib.th_last_error().decode())
        self._n = 0
        self._ts = None
        self._ts_n = 0
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.th_destroy(self.h)
            self.h = None
        if self._ts is not None:
            self.lib.th_free_tandem_seq(self._ts, self._ts_n)
            self._ts = None
        if getattr(self, "para", None):
            self.lib.th_free_para(self.para)
            self.para = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("tidehunter_amd: " + self.lib.th_last_error().decode())

    def upload(self, seqs):
        n = len(seqs)
        arr = (C.c_char_p * n)(*seqs)
        lens = (C.c_int * n)(*[len(s) for s in seqs])
        self._keep = (arr, lens, seqs)
        self._n = n
        if self._ts is None or self._ts_n < n:
            if self._ts is not None:
                self.lib.th_free_tandem_seq(self._ts, self._ts_n)
            self._ts = self.lib.th_alloc_tandem_seq(max(n, 1))
            self._ts_n = max(n, 1)
        self._check(self.lib.th_batch_upload(self.h, arr, lens, n))

    def run(self):
        self._check(self.lib.th_batch_run(self.h))

    def collect_native(self):
        """th_batch_collect() only: fills the th_tandem_seq_t slots (host arithmetic in C); no Python objects"""
        self._check(self.lib.th_batch_collect(self.h, self._ts))

    def records(self):
        """Python view of the slots filled by collect_native()"""
        return [self._records(i) for i in range(self._n)]

    def collect(self):
        self.collect_native()
        return self.records()

    KERNELS = ("th_k_encode+th_k_hits", "th_k_chain_dp", "th_k_chain_extract", "th_k_partition", "th_k_poa", "th_k_align")

    def timing(self):
        """device ms of the last run() per kernel (HIP events on the library's stream)"""
        a = (C.c_double * 8)()
        n = self.lib.th_batch_timing(self.h, a, 8)
        return {self.KERNELS[i]: a[i] for i in range(n)}

    def launches(self):
        """concurrent parts the last uploaded batch runs as (1, or 2 for batches of TH_SUBBATCH_MIN = 16384 reads and more)"""
        return self.lib.th_batch_launches(self.h)

    def run_steps(self, steps):
        """th_batch_run_steps(): the resident batch `steps` times, record arithmetic of pass k under the kernels of pass k+1; the slots hold
        the last pass's records afterwards (records()).  Returns the device ms per kernel summed over the passes."""
        a = (C.c_double * 8)()
        self._check(self.lib.th_batch_run_steps(self.h, steps, self._ts, a))
        return {self.KERNELS[i]: a[i] for i in range(6)}

    def stats(self):
        """cumulative counters of the handle's recovery paths (th_batch_stats): consensus jobs left without a consensus, POA jobs rerun,
        reruns of stages 2-3 / the job tables, batches, consensus jobs"""
        v = (C.c_longlong * 8)()
        self.lib.th_batch_stats(self.h, v, 8)
        return dict(zip(("jobs_without_consensus", "poa_jobs_rerun", "stage_reruns", "batches", "jobs"), [int(x) for x in v[:5]]))

    def syncs(self):
        """host synchronisations (stream drains) of the last run()"""
        return self.lib.th_batch_syncs(self.h)

    def prepare_batches(self, batches):
        """marshals a list of batches for run_prepared(): the ctypes arrays and one tandem_seq_t array per batch (what a C caller has at hand
        before it calls th_process_batches())"""
        nb = len(batches)
        uniq = {}
        for b in batches:                      # (the bench passes the same batch several times: marshal it once)
            if id(b) not in uniq:
                n = len(b)
                uniq[id(b)] = ((C.c_char_p * n)(*b), (C.c_int * n)(*[len(s) for s in b]), n)
        outs = [self.lib.th_alloc_tandem_seq(max(len(b), 1)) for b in batches]
        pp, ip = C.POINTER(C.c_char_p), C.POINTER(C.c_int)
        a_seqs = (pp * nb)(*[C.cast(uniq[id(b)][0], pp) for b in batches])
        a_lens = (ip * nb)(*[C.cast(uniq[id(b)][1], ip) for b in batches])
        a_n = (C.c_int * nb)(*[uniq[id(b)][2] for b in batches])
        a_out = (C.POINTER(TandemSeq) * nb)(*outs)
        return dict(nb=nb, seqs=a_seqs, lens=a_lens, n=a_n, out=a_out, outs=outs, sizes=[max(len(b), 1) for b in batches], keep=(uniq, batches))

    def run_prepared(self, ctx):
        """th_process_batches() on prepared batches: upload, kernels and record arithmetic of neighbouring batches overlapped"""
        self._check(self.lib.th_process_batches(self.h, ctx["nb"], ctx["seqs"], ctx["lens"], ctx["n"], ctx["out"]))

    def free_prepared(self, ctx):
        for o, n in zip(ctx["outs"], ctx["sizes"]):
            self.lib.th_free_tandem_seq(o, n)
        ctx["outs"] = []

    def process_batches(self, batches):
        """th_process_batches(): the batches back to back through the upload / kernels / record-arithmetic pipeline; the result slots
        stay with the library (use process() to look at records) -- this entry exists to time the pipelined seam"""
        ctx = self.prepare_batches(batches)
        try:
            self.run_prepared(ctx)
        finally:
            self.free_prepared(ctx)

    def process(self, seqs):
        """tidehunter_core() on every read of the batch; returns a list (per read) of record dicts."""
        self.upload(seqs)
        self.run()
        return self.collect()

    def _records(self, i):
        t = self._ts[i]
        out, cs = [], 0
        only_unit = self.para.contents.only_unit
        for c in range(t.cons_n):
            sub = [t.sub_pos[c][j] for j in range(t.pos_n[c])]
            if only_unit:
                out.append(dict(sub_pos=sub))
                continue
            L = t.cons_len[c]
            ks, kq = t.cons_seq.contents.seq, t.cons_seq.contents.qual
            seq = C.string_at(C.addressof(ks.s.contents) + cs, L)
            qual = C.string_at(C.addressof(kq.s.contents) + cs, L) if kq.s and kq.l >= cs + L else None
            out.append(dict(seq=seq, qual=qual, cons_len=L, start=t.cons_start[c], end=t.cons_end[c], copy_num=t.copy_num[c],
                            ave_match=t.ave_match[c], full_length=int(t.full_length[c]), sub_pos=sub))
            cs += L
        return out

    def format_output(self, names):
        """th_format_output(): the text mini_tandem_output() would print for the resident batch's records (names: one bytes per read);
        the slots are reset, as th_output() does"""
        n = self._n
        assert len(names) == n
        arr_n = (C.c_char_p * n)(*names)
        ln = C.c_size_t(0)
        self.lib.th_format_output.restype = C.c_void_p
        self.lib.th_format_output.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(TandemSeq), C.POINTER(Para), C.POINTER(C.c_size_t)]
        self.lib.th_free_buffer.argtypes = [C.c_void_p]
        p = self.lib.th_format_output(n, arr_n, self._keep[0], self._keep[1], self._ts, self.para, C.byref(ln))
        try:
            return C.string_at(p, ln.value)
        finally:
            self.lib.th_free_buffer(p)

    def debug_stage(self, what, nbytes, dtype):
        import numpy as np
        buf = np.zeros(max(1, nbytes // np.dtype(dtype).itemsize + 8), dtype=dtype)
        got = self.lib.th_debug_stage(self.h, what, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        if got < 0:
            raise RuntimeError("debug_stage(%d) failed" % what)
        return buf[: got // np.dtype(dtype).itemsize]


def synth_read(lib, seed, index, period, copies, err):
    cap = int(period * copies * (1 + err) + 64) + period
    buf = C.create_string_buffer(cap)
    n = lib.th_synth_read(seed, index, period, copies, err, buf, cap)
    assert n <= cap
    return buf.raw[:n]


C5_ADAPTER_SEED, C5_FIVE_LEN, C5_THREE_LEN = 55, 84, 25


def synth_adapters(lib, seed=C5_ADAPTER_SEED):
    """(5' adapter, 3' adapter) of config 5"""
    out = []
    for which, n in ((0, C5_FIVE_LEN), (1, C5_THREE_LEN)):
        buf = C.create_string_buffer(n + 1)
        lib.th_synth_adapter(seed, which, buf, n)
        out.append(buf.raw[:n])
    return tuple(out)


def synth_read_c5(lib, seed, index, adapters, err=0.10):
    """config 5 read `index`: period {200,500,1000,2000,5000}[index % 5], copies max(3, 10000 // period), both strands"""
    cap = 20000
    buf = C.create_string_buffer(cap)
    n = lib.th_synth_read_c5(seed, index, adapters[0], len(adapters[0]), adapters[1], len(adapters[1]), err, buf, cap)
    assert 0 < n <= cap
    return buf.raw[:n]
