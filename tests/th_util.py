"""Shared helpers for the parity tests (CPU emulation and GPU)."""
import os
import subprocess

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
EMUL_SO = os.path.join(ROOT, "tests", "cpu_emul", "libth_emul.so")


def effective_cpus():
    """usable CPUs (affinity capped by the cgroup quota): see bench.effective_cpus"""
    import bench
    return bench.effective_cpus()


def synth_lib():
    """the product library, for its deterministic read generator only (th_synth_read needs no GPU)"""
    import tidehunter_amd as TH
    return TH.load_library()


def synth_read(lib, seed, index, period, copies, err):
    import tidehunter_amd as TH
    return TH.synth_read(lib, seed, index, period, copies, err)


def emul_lib_path():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpu_emul")], check=True, stdout=subprocess.DEVNULL)
    return EMUL_SO


def read_fastx(path):
    """tiny FASTA/FASTQ reader for test inputs -> list of (name, seq bytes)"""
    out, name, seq, fq = [], None, [], False
    lines = open(path, "rb").read().split(b"\n")
    i = 0
    while i < len(lines):
        l = lines[i].rstrip(b"\r")
        if l.startswith(b">"):
            if name is not None:
                out.append((name, b"".join(seq)))
            name, seq = l[1:].split()[0], []
        elif l.startswith(b"@") and i + 3 < len(lines) and lines[i + 2].startswith(b"+"):
            if name is not None:
                out.append((name, b"".join(seq)))
            out.append((l[1:].split()[0], lines[i + 1].rstrip(b"\r")))
            name, seq = None, []
            i += 3
        elif l:
            seq.append(l)
        i += 1
    if name is not None:
        out.append((name, b"".join(seq)))
    return out


def oracle_records(seqs, para):
    """tho_core() on every read; the calls release the GIL, so the host cores of the box work in parallel"""
    from concurrent.futures import ThreadPoolExecutor
    if len(seqs) < 8:
        return [O.core(s, para) for s in seqs]
    with ThreadPoolExecutor(min(64, effective_cpus())) as ex:
        return list(ex.map(lambda s: O.core(s, para), seqs))


def assert_records_equal(seqs, got, para, label="", want=None):
    """got: TideHunter.process() output; compared field by field with the oracle's tho_core() (want: its records when the caller already has them)."""
    n = 0
    if want is None:
        want = oracle_records(seqs, para)
    assert len(want) == len(got) == len(seqs), (label, len(want), len(got), len(seqs))
    for idx, (s, recs) in enumerate(zip(seqs, got)):
        exp = want[idx]
        assert len(exp) == len(recs), "%s read %d: %d records, oracle %d" % (label, idx, len(recs), len(exp))
        for a, b in zip(exp, recs):
            assert a["sub_pos"] == b["sub_pos"], (label, idx)
            if para.only_unit:
                continue
            assert a["seq"] == b["seq"], (label, idx, a["seq"], b["seq"])
            assert (a["start"], a["end"], a["cons_len"], a["full_length"]) == (b["start"], b["end"], b["cons_len"], b["full_length"]), (label, idx)
            # doubles come from the same integer evidence and the same expressions: bit-identical
            assert a["copy_num"] == b["copy_num"] and a["ave_match"] == b["ave_match"], (label, idx)
            if a["qual"] is not None or b["qual"] is not None:
                assert a["qual"] == b["qual"], (label, idx)
            n += 1
    return n


def stage_compare(th, seqs, para):
    """hits / chain DP / chains / par_pos of the resident batch against the oracle's stage dumps."""
    T = sum(len(s) for s in seqs)
    roff = th.debug_stage(8, 4 * (len(seqs) + 1), np.uint32)
    hn = th.debug_stage(1, 4 * len(seqs), np.int32)
    he, hp = th.debug_stage(2, 4 * T, np.int32), th.debug_stage(3, 4 * T, np.int32)
    sc, fr = th.debug_stage(4, 4 * T, np.int32), th.debug_stage(5, 4 * T, np.int32)
    rec = th.debug_stage(6, 32 * (8 * len(seqs) + 4096), np.int32).reshape(-1, 8)
    cells = th.debug_stage(7, 4 * T, np.int32)
    parc = th.debug_stage(9, 8 * T + 1024, np.int32)
    bseq = th.debug_stage(0, T, np.uint8)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(64, effective_cpus())) as ex:
        all_st = list(ex.map(lambda s: O.stages(s, para), seqs))
    for r, s in enumerate(seqs):
        st = all_st[r]
        o, n = int(roff[r]), int(hn[r])
        assert np.array_equal(bseq[o:o + len(s)], st["bseq"]), r
        hits = (he[o:o + n].astype(np.uint64) << np.uint64(32)) | hp[o:o + n].astype(np.uint64)
        assert np.array_equal(hits, st["hits"]), "hits differ, read %d" % r
        if n >= 2:
            assert np.array_equal(sc[o:o + n], st["score"]), "chain scores differ, read %d" % r
            assert np.array_equal(fr[o:o + n], st["frm"]), "chain predecessors differ, read %d" % r
        mine = [x for x in rec if x[0] == r]
        assert len(mine) == len(st["chains"]), "chain count differs, read %d" % r
        for m, c in zip(mine, st["chains"]):
            assert list(cells[m[1]:m[1] + m[2]]) == c["cells"], r
            assert (m[3], m[4]) == (c["est_start"], c["est_period"]), r
            assert list(parc[m[5]:m[5] + m[6]]) == c["par_pos"], "par_pos differs, read %d" % r


def config5_reads(tmpdir, n, seed=5, err=0.10, periods=(200, 500, 1000)):
    """SURVEY 8(d) config 5 shape: unit = 5' adapter (84 nt) + random insert + revcomp(3' adapter, 25 nt), tandemly repeated
    with substitutions / insertions / deletions; half of the reads reverse-complemented.  Writes 5p.fa / 3p.fa into tmpdir and
    returns (reads, (five_fn, three_fn))."""
    rng = np.random.default_rng(seed)
    rnd = lambda k: bytes(b"ACGT"[i] for i in rng.integers(0, 4, k))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = lambda s: s.translate(comp)[::-1]
    five, three = rnd(84), rnd(25)
    fns = (os.path.join(str(tmpdir), "5p.fa"), os.path.join(str(tmpdir), "3p.fa"))
    open(fns[0], "wb").write(b">5p\n" + five + b"\n")
    open(fns[1], "wb").write(b">3p\n" + three + b"\n")
    reads = []
    for r in range(n):
        period = periods[r % len(periods)]
        unit = five + rnd(period - 84 - 25) + rc(three)
        copies = max(3, 3000 // period)
        out = bytearray(rnd(int(rng.integers(0, 40))))           # a little flank
        for b in unit * copies:
            x = rng.random()
            if x < err / 3:
                out.append(b"ACGT"[(b"ACGT".index(b) + 1 + int(rng.integers(0, 3))) & 3])
            elif x < 2 * err / 3:
                out.append(b); out.append(b"ACGT"[int(rng.integers(0, 4))])
            elif x < err:
                pass
            else:
                out.append(b)
        s = bytes(out)
        reads.append(rc(s) if r & 1 else s)
    return reads, fns


def write_fasta(path, seqs, prefix="r"):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">%s%d\n" % (prefix.encode(), i) + s + b"\n")


def write_fastq_gz(path, seqs, prefix="r", line=0):
    """gzipped FASTQ; line > 0 wraps sequence and quality lines (multi-line FASTQ, which the kseq grammar allows)"""
    import gzip
    with gzip.open(path, "wb", compresslevel=1) as f:
        for i, s in enumerate(seqs):
            q = bytes(33 + (j * 7 + i) % 40 for j in range(len(s)))
            if line > 0:
                wrap = lambda b: b"\n".join(b[k:k + line] for k in range(0, len(b), line)) if b else b""
                f.write(b"@%s%d some comment\n" % (prefix.encode(), i) + wrap(s) + b"\n+\n" + wrap(q) + b"\n")
            else:
                f.write(b"@%s%d\n" % (prefix.encode(), i) + s + b"\n+\n" + q + b"\n")


def config5_spec(tmpdir, n, seed=5):
    """SURVEY.md 8(d) config 5 AT SPEC: periods {200, 500, 1000, 2000, 5000} by read index, copies max(3, 10000 // period),
    10 % error, both strands, 84-nt 5' / 25-nt 3' adapters from seed 55 written to 5p.fa / 3p.fa."""
    import tidehunter_amd as TH
    lib = synth_lib()
    ad = TH.synth_adapters(lib)
    fns = (os.path.join(str(tmpdir), "5p.fa"), os.path.join(str(tmpdir), "3p.fa"))
    open(fns[0], "wb").write(b">5p\n" + ad[0] + b"\n")
    open(fns[1], "wb").write(b">3p\n" + ad[1] + b"\n")
    return [TH.synth_read_c5(lib, seed, i, ad) for i in range(n)], fns


def write_bgzf(path, data: bytes, block=0xff00, level=1):
    """block-gzipped file (bgzip / BGZF: gzip members <= 64 KiB with their size in a 'BC' extra subfield, then the empty EOF member)"""
    import struct
    import zlib
    with open(path, "wb") as f:
        for off in list(range(0, len(data), block)) + [None]:
            chunk = b"" if off is None else data[off:off + block]
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            bsize = 12 + 6 + len(comp) + 8
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            f.write(comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
