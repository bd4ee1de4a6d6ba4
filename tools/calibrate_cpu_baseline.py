#!/usr/bin/env python3
"""Build container only (needs /root/reference): times the parts of the CPU baseline that CAN be compared with the reference's own code --
oracle/_ref/libthref.so = the reference's ksw2 (SSE), edlib and seq.c compiled where they lie -- against the oracle's scalar restatements
(oracle/libtho.so) on IDENTICAL inputs of the headline shape, one thread each, and writes profiles/r04_cpu_calibration.json.

Why: bench.py's `cpu_baseline` is the oracle port (kind "port"), not the reference binary (abPOA is absent: the reference cannot be built).
The ratios below say how far the port's per-read cost may be read as the reference's: per unit-vs-consensus alignment and per flank
extension (src/gen_cons.c:209-223 -> src/ksw2_align.c:117-173), per adapter search (src/edlib_align.c:73-85) and per encoded base
(src/seq.c:77-86).  Stages 1-3 (tandem_hit.c / tandem_chain.c / partition.c) and abPOA cannot be timed from the reference here; SURVEY.md
section 6 holds the survey's own stage 1-3 figures (9.2 ms per read per core at this shape)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


def mutate(rng, s, rate):
    out = []
    for b in s:
        r = rng.random()
        if r < rate / 3:
            out.append((int(b) + 1 + int(rng.integers(0, 3))) & 3)
        elif r < 2 * rate / 3:
            out.append(int(b)); out.append(int(rng.integers(0, 4)))
        elif r < rate:
            continue
        else:
            out.append(int(b))
    return np.array(out, dtype=np.uint8)


def timed(fn, reps):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    ref, lib = O.load_ref(), O.load()
    if ref is None:
        sys.exit("oracle/_ref/libthref.so is not built (needs /root/reference)")
    rng = np.random.default_rng(4)
    u8p = C.POINTER(C.c_uint8)
    pairs = []
    for _ in range(24):                        # unit against consensus at the headline shape: ~5 % each way
        cons = rng.integers(0, 4, 1000).astype(np.uint8)
        pairs.append((np.ascontiguousarray(mutate(rng, cons, 0.10)), cons))
    res = {}

    def both(name, f_ref, f_port, reps, unit):
        a, b = timed(f_ref, reps), timed(f_port, reps)
        res[name] = {"reference_s": a, "port_s": b, "port_over_reference": b / a, "per": unit}
        print("%-34s reference %.3e s   port %.3e s   port / reference %.2f   (per %s)" % (name, a, b, b / a, unit))

    def run_pairs(fn):
        def go():
            for q, t in pairs:
                fn(q.ctypes.data_as(u8p), len(q), t.ctypes.data_as(u8p), len(t))
        return go
    mq, mt = C.c_int(), C.c_int()
    both("ksw2_global 1000 x 1000", run_pairs(ref.ksw2_global), run_pairs(lib.tho_ksw2_global), 3, "%d alignments" % len(pairs))
    both("ksw2_right_ext 1000 x 1000", run_pairs(lambda a, b, c, d: ref.ksw2_right_ext(a, b, c, d, C.byref(mq), C.byref(mt))),
         run_pairs(lambda a, b, c, d: lib.tho_ksw2_right_ext(a, b, c, d, C.byref(mq), C.byref(mt))), 3, "%d extensions" % len(pairs))
    both("ksw2_left_ext 1000 x 1000", run_pairs(lambda a, b, c, d: ref.ksw2_left_ext(a, b, c, d, C.byref(mq), C.byref(mt))),
         run_pairs(lambda a, b, c, d: lib.tho_ksw2_left_ext(a, b, c, d, C.byref(mq), C.byref(mt))), 3, "%d extensions" % len(pairs))
    q = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 84)); t = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 2000))
    s, e = C.c_int(), C.c_int()
    both("edlib_align_HW 84 in 2000", lambda: [ref.edlib_align_HW(q, 84, t, 2000, C.byref(s), C.byref(e), 16) for _ in range(200)],
         lambda: [lib.tho_edlib_hw(q, 84, t, 2000, C.byref(s), C.byref(e), 16) for _ in range(200)], 3, "200 searches")
    read = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 10000)); outb = np.zeros(10000, dtype=np.uint8)
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]; ref.get_bseq.restype = C.c_void_p

    def ref_enc():
        for _ in range(200):
            libc.free(ref.get_bseq(read, 10000))
    both("get_bseq 10 kb", ref_enc, lambda: [lib.tho_encode(read, 10000, outb.ctypes.data_as(u8p)) for _ in range(200)], 3, "200 reads")
    # what the ratios mean for the per-read figure of the port (config 3: 9 unit alignments + 2 extensions per read)
    n_u, n_x = 9, 2
    port_ksw = (res["ksw2_global 1000 x 1000"]["port_s"] * n_u + (res["ksw2_right_ext 1000 x 1000"]["port_s"] + res["ksw2_left_ext 1000 x 1000"]["port_s"]) * n_x / 2) / len(pairs)
    ref_ksw = (res["ksw2_global 1000 x 1000"]["reference_s"] * n_u + (res["ksw2_right_ext 1000 x 1000"]["reference_s"] + res["ksw2_left_ext 1000 x 1000"]["reference_s"]) * n_x / 2) / len(pairs)
    out = {"note": "one thread, this container (build box), identical inputs; reference = oracle/_ref/libthref.so (the reference's own ksw2 SSE / edlib / seq.c), port = oracle/libtho.so",
           "functions": res,
           "per_read_ksw2_ms_config3": {"port": 1000 * port_ksw, "reference": 1000 * ref_ksw,
                                        "what": "9 unit-vs-consensus alignments + 2 flank extensions of ~1000 x 1000 per read (src/gen_cons.c:209-223)"},
           "reading": "the port's ksw2 share of a headline read costs %.1f ms where the reference's SSE code takes %.1f ms (SURVEY.md section 6 measured ~10.5 ms per read for it inside the "
                      "reference); stages 1-3 and the POA of the port have no reference timing here (abPOA absent), so cpu_baseline stays kind 'port' and the GPU / CPU ratio of the bench line is a "
                      "ratio against the port, not against TideHunter" % (1000 * port_ksw, 1000 * ref_ksw)}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_cpu_calibration.json"), "w"), indent=1)
    print(out["reading"])


if __name__ == "__main__":
    main()
