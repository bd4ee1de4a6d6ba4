#!/bin/bash
# GPU box (or anywhere): the FASTA reader alone on N x the 100 000-read headline set in /dev/shm -- line index (helper threads) against
# the parser's own memchr, wall and CPU seconds.   usage: tools/reader_rate.sh [copies=10]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python3 - <<PY
import sys, os, time, ctypes as C
sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
fn = "/dev/shm/th_reader_rate.fa"
reads = [TH.synth_read(lib, 1, i, 1000, 10, 0.10) for i in range(100000)]
with open(fn, "wb") as f:
    for c in range(${1:-10}):
        f.write(b"".join(b">syn_%d\n" % (c * 100000 + i) + s + b"\n" for i, s in enumerate(reads)))
sz = os.path.getsize(fn)
for env in ({}, {"TH_READER_NOINDEX": "1"}, {"TH_READER_THREADS": "1"}, {"TH_READER_THREADS": "2"}, {"TH_READER_THREADS": "6"}, {}, {"TH_READER_NOINDEX": "1"}):
    os.environ.update(env)
    t0, c0 = time.time(), time.process_time()
    n = lib.th_parse_file(fn.encode(), None, None)
    dt, cpu = time.time() - t0, time.process_time() - c0
    for k in env: del os.environ[k]
    print("%-32s %d reads  %.3f s wall  %.2f GB/s  %.3f s CPU" % (env or "index, default helpers", n, dt, sz / dt / 1e9, cpu))
os.unlink(fn)
PY
