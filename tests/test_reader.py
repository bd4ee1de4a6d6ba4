"""The FASTA/FASTQ reader alone (th_parse_file: no device involved): the mmap'ed zero-copy path, the parallel BGZF path and the
zlib stream path must see the same records (kseq grammar, src/kseq.h:176-215: multi-line records, CRLF, FASTQ with '@' in the
quality line, comments after the name, empty lines), and the mmap path must be fast enough to feed several GPUs."""
import ctypes as C
import gzip
import os
import time

import th_util as U


def _parse(lib, fn, digest=True, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        nb, d = C.c_longlong(), C.c_ulonglong()
        t0 = time.time()
        n = lib.th_parse_file(str(fn).encode(), C.byref(nb), C.byref(d) if digest else None)
        return n, nb.value, d.value if digest else None, time.time() - t0
    finally:
        for k in (env or {}):
            del os.environ[k]


def test_reader_paths_agree_on_awkward_input(tmp_path):
    lib = U.synth_lib()
    seqs = [U.synth_read(lib, 4, i, 90 + i, 5, 0.1) for i in range(40)]
    rec = []
    for i, s in enumerate(seqs):
        if i % 4 == 0:      # multi-line FASTA with CRLF and a comment
            rec.append(b">r%d some comment\r\n" % i + b"\r\n".join(s[k:k + 60] for k in range(0, len(s), 60)) + b"\r\n")
        elif i % 4 == 1:    # one-line FASTA, empty line after it
            rec.append(b">r%d\n" % i + s + b"\n\n")
        elif i % 4 == 2:    # FASTQ whose quality line starts with '@' and contains '>' and '+'
            rec.append(b"@r%d\n" % i + s + b"\n+\n@" + b">+" * ((len(s) - 1) // 2) + b"I" * ((len(s) - 1) % 2) + b"\n")
        else:               # multi-line FASTQ
            q = bytes(33 + (k % 40) for k in range(len(s)))
            rec.append(b"@r%d\tx\n" % i + s[:50] + b"\n" + s[50:] + b"\n+r%d\n" % i + q[:70] + b"\n" + q[70:] + b"\n")
    raw = b"garbage before the first record\n" + b"".join(rec)
    plain, gz, bgz = tmp_path / "in.fx", tmp_path / "in.fx.gz", tmp_path / "in.fx.bgz"
    plain.write_bytes(raw)
    with gzip.open(gz, "wb") as f:
        f.write(raw)
    U.write_bgzf(bgz, raw, block=700)
    want = (len(seqs), sum(len(s) for s in seqs))
    got = [_parse(lib, plain), _parse(lib, plain, env={"TH_READER_STREAM": "1"}), _parse(lib, gz), _parse(lib, bgz),
           _parse(lib, bgz, env={"TH_READER_THREADS": "1"}), _parse(lib, bgz, env={"TH_READER_STREAM": "1"})]
    assert all(g[:2] == want for g in got), [g[:2] for g in got]
    assert len({g[2] for g in got}) == 1
    # "-" = standard input (the reference reads stdin through the same gzread, src/main.c:378), plain and gzipped
    import subprocess
    import sys
    child = ("import sys, ctypes as C; sys.path.insert(0, %r); import tidehunter_amd as TH; lib = TH.load_library(); "
             "nb = C.c_longlong(); d = C.c_ulonglong(); n = lib.th_parse_file(b'-', C.byref(nb), C.byref(d)); print(n, nb.value, d.value)" % U.ROOT)
    for src in (plain, gz):
        with open(src, "rb") as f:
            out = subprocess.run([sys.executable, "-c", child], stdin=f, capture_output=True, text=True, check=True).stdout.split()
        assert (int(out[0]), int(out[1]), int(out[2])) == (want[0], want[1], got[0][2])
    assert lib.th_parse_file(str(tmp_path / "missing.fa").encode(), None, None) == -1
    (tmp_path / "empty.fa").write_bytes(b"")
    assert _parse(lib, tmp_path / "empty.fa")[:2] == (0, 0)


def test_bgzf_group_boundary_and_damage(tmp_path):
    """exactly 512 data members (one reader group) followed by the empty end-of-file member: the second group holds only that member
    and must not be reported as corrupt; a truncated file IS reported (-2), not parsed as if it were complete"""
    lib = U.synth_lib()
    seqs = [U.synth_read(lib, 6, i, 50, 4, 0.05) for i in range(300)]
    raw = b"".join(b">r%d\n" % i + s + b"\n" for i, s in enumerate(seqs))
    block = -(-len(raw) // 512)
    assert -(-len(raw) // block) == 512
    bgz = tmp_path / "g512.fa.bgz"
    U.write_bgzf(bgz, raw, block=block)
    n, nb, _, _ = _parse(lib, bgz)
    assert (n, nb) == (len(seqs), sum(len(s) for s in seqs))
    cut = tmp_path / "cut.fa.bgz"
    cut.write_bytes(bgz.read_bytes()[: os.path.getsize(bgz) // 2 + 7])        # (mid-member)
    assert _parse(lib, cut)[0] == -2


def test_mmap_reader_rate(tmp_path):
    """single-line 10 kb reads (the headline shape) from the page cache: several GB/s on one thread (7 GB/s measured on the build
    container, where the zlib stream path reaches 2.3 GB/s on the same file) -- the bar of SURVEY 8(f).2 is ~7 GB/s for 8 GPUs"""
    lib = U.synth_lib()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    fn = os.path.join(d, "th_reader_rate_%d.fa" % os.getpid())
    try:
        U.write_fasta(fn, [U.synth_read(lib, 1, i, 1000, 10, 0.10) for i in range(6000)])
        size = os.path.getsize(fn)
        best = min(_parse(lib, fn, digest=False)[3] for _ in range(5))
        stream = min(_parse(lib, fn, digest=False, env={"TH_READER_STREAM": "1"})[3] for _ in range(3))
        print("mmap reader %.2f GB/s, stream reader %.2f GB/s" % (size / best / 1e9, size / stream / 1e9))
        assert size / best / 1e9 > 2.0 and best < stream
    finally:
        if os.path.exists(fn):
            os.unlink(fn)


def test_line_index_agrees_with_the_plain_scan(tmp_path):
    """mapped inputs of 16 MiB and more are parsed through the line index (helper threads leave the newline offsets of 4 MiB chunks,
    the grammar walks them on one thread): same records as the parser's own memchr (TH_READER_NOINDEX=1) and as the zlib stream path on
    a file that mixes one-line 10 kb reads, multi-line records, FASTQ whose quality lines start with '@', CRLF, a line longer than a
    chunk and a stretch of one-byte lines (a dense chunk, which is not indexed)"""
    lib = U.synth_lib()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    fn = os.path.join(d, "th_reader_lidx_%d.fx" % os.getpid())
    long_read = U.synth_read(lib, 2, 0, 1000, 10, 0.10)
    try:
        n_rec, n_bases = 0, 0
        with open(fn, "wb") as f:
            for i in range(2600):
                s = U.synth_read(lib, 9, i, 400 + (i % 7) * 100, 8, 0.1) if i % 50 else long_read
                if i % 5 == 0:
                    f.write(b">m%d c\r\n" % i + b"\r\n".join(s[k:k + 61] for k in range(0, len(s), 61)) + b"\r\n")
                elif i % 5 == 1:
                    f.write(b"@q%d\n" % i + s + b"\n+\n@" + b"+>" * ((len(s) - 1) // 2) + b"@" * ((len(s) - 1) % 2) + b"\n")
                else:
                    f.write(b">s%d\n" % i + s + b"\n")
                n_rec += 1; n_bases += len(s)
                if i == 700:      # a sequence line longer than a chunk
                    big = long_read * 500
                    f.write(b">big\n" + big + b"\n"); n_rec += 1; n_bases += len(big)
                if i == 1500:     # a dense stretch: a multi-line record of one base per line, then empty lines
                    f.write(b">dense\n" + b"\n".join(bytes([c]) for c in (long_read * 30)) + b"\n" + b"\n" * 300000); n_rec += 1; n_bases += len(long_read) * 30
        assert os.path.getsize(fn) > (20 << 20)
        a = _parse(lib, fn)
        b = _parse(lib, fn, env={"TH_READER_NOINDEX": "1"})
        c = _parse(lib, fn, env={"TH_READER_STREAM": "1"})
        e = _parse(lib, fn, env={"TH_READER_THREADS": "1"})
        assert a[:2] == (n_rec, n_bases), (a[:2], n_rec, n_bases)
        assert a[:3] == b[:3] == c[:3] == e[:3]
    finally:
        if os.path.exists(fn):
            os.unlink(fn)
