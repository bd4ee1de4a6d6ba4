"""GPU parity at scale, reproduced by the driver (`pytest -m gpu`): what tools/parity_scale.sh did by hand in round 1.

  * byte identity of the command line (reader -> batch seam -> writer) against the CPU oracle's command line on thousands
    of synthetic reads per BASELINE.json config shape and several option sets;
  * config 5 AT SPEC: periods {200, 500, 1000, 2000, 5000}, copies max(3, 10000 // period), both strands,
    5'/3' adapters, gzipped (multi-line) FASTQ input through the command line, with and without -F / -s;
  * the HIP POA kernel against the INDEPENDENT exact POA (oracle/tho_poa_exact.c: full matrix, Kahn order) on >= 2000 jobs
    per config shape when the host has the cores for it.
Sample sizes scale with the CPUs the container may use (the oracle costs ~35-70 ms per headline read per core); with the 16-CPU
quota of the GPU boxes they reach the sizes quoted above (4096 / 4096 / 512 reads, 2048 / 2048 / 256 POA jobs).  Everything POA touches is "GPU == oracle", not "== reference" (abPOA is absent).
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

import oracle_lib as O
import th_util as U
import tidehunter_amd as TH

pytestmark = pytest.mark.gpu

CORES = U.effective_cpus()
EXE = os.path.join(U.ROOT, "tidehunter_amd", "TideHunter-amd")
ORACLE = os.path.join(O.ODIR, "tho_oracle")


def _n(full, per_core):
    return int(min(full, max(48, per_core * CORES)))


def _run(cmd):
    return subprocess.run(cmd, check=True, capture_output=True).stdout


def _both(tmp_path, opts, infile):
    O.load()
    r = subprocess.run([EXE, *opts, str(infile)], check=True, capture_output=True)
    # a consensus job the POA kernels could not serve ends WITHOUT a consensus behind a warning on stderr (the reference has no such failure mode,
    # src/abpoa_cons.c:86-87): none may occur at scale
    assert b"could not be aligned" not in r.stderr, r.stderr[-400:]
    gpu = r.stdout
    cpu = _run([ORACLE, "-t", str(CORES), *opts, str(infile)])
    return gpu, cpu


@pytest.fixture(scope="module")
def shape_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("scale")
    lib = TH.load_library()
    files = {}
    for name, (n, period, copies, err, seed) in dict(C3=(_n(4096, 256), 1000, 10, 0.10, 1), C2=(_n(4096, 256), 200, 20, 0.10, 2),
                                                    C4=(_n(512, 32), 5000, 4, 0.15, 3)).items():
        fn = d / (name + ".fa")
        U.write_fasta(fn, [TH.synth_read(lib, seed, i, period, copies, err) for i in range(n)], prefix="syn_%s_" % name)
        files[name] = (fn, n)
    return files


@pytest.mark.parametrize("shape,opts", [
    ("C3", []), ("C3", ["-f", "2"]), ("C3", ["-f", "4", "-c", "3"]),
    ("C2", []), ("C2", ["-u", "-f", "2"]), ("C2", ["-k", "10", "-w", "5"]), ("C2", ["-H"]),
    ("C4", []), ("C4", ["-f", "2", "-l"]),
])
def test_gpu_cli_byte_identity_at_scale(tmp_path, shape_files, shape, opts):
    fn, n = shape_files[shape]
    gpu, cpu = _both(tmp_path, opts, fn)
    assert gpu == cpu, "%s %r: command-line output differs from the oracle's" % (shape, opts)
    assert gpu.count(b"\n") >= n // 2


def test_gpu_config5_at_spec_gz_fastq(tmp_path):
    """config 5 at its SURVEY 8(d) definition, >= 200 reads, gz FASTQ through the command line; all five period classes with
    adapters (the 2000 / 5000 classes are the ones that reach th_k_poa<8192,int> and th_k_align<24576,6>)"""
    n = max(200, _n(1000, 20))
    n -= n % 10
    seqs, ad = U.config5_spec(tmp_path, n)
    assert {len(s) // 4000 for s in seqs} >= {2, 3}                      # ~10 kb reads and the ~15 kb reads of the 5000 class
    fq = tmp_path / "c5.fq.gz"
    U.write_fastq_gz(fq, seqs, prefix="c5_", line=0)
    fq_wrapped = tmp_path / "c5_wrapped.fq.gz"
    U.write_fastq_gz(fq_wrapped, seqs[:40], prefix="c5_", line=70)      # multi-line FASTQ records
    adopt = ["-5", ad[0], "-3", ad[1]]
    for opts in (adopt, adopt + ["-F", "-f", "2"], adopt + ["-F", "-s", "-f", "4"]):
        gpu, cpu = _both(tmp_path, opts, fq)
        assert gpu == cpu, "config 5 %r: output differs from the oracle's" % (opts[4:],)
    gpu, cpu = _both(tmp_path, adopt, fq_wrapped)
    assert gpu == cpu and gpu.count(b">") >= 30
    # the same records block-gzipped (bgzip: inflated by the reader's thread pool) and uncompressed (mmap'ed, zero-copy)
    import gzip
    raw = gzip.open(fq).read()
    bgz, plain = tmp_path / "c5.fq.bgz", tmp_path / "c5.fq"
    U.write_bgzf(bgz, raw); plain.write_bytes(raw)
    want = _run([EXE, *adopt, str(fq)])
    assert _run([EXE, *adopt, str(bgz)]) == want and _run([EXE, *adopt, str(plain)]) == want
    # the records themselves, through the batch seam: every period class yields full-length consensuses on both strands
    para = O.default_para()
    O.load().tho_para_load_adapters(para, ad[0].encode(), ad[1].encode())
    t = TH.TideHunter(device=0, adapters=ad)
    got = t.process(seqs)
    assert U.assert_records_equal(seqs, got, para, "config 5 at spec") >= n * 9 // 10
    by_class = {}
    for i, rr in enumerate(got):
        for r in rr:
            if r["full_length"] > 0:
                by_class.setdefault(i % 5, set()).add(r["full_length"])
    assert all(by_class.get(c) == {1, 2} for c in range(5)), by_class
    t.close()


def test_gpu_config5_sharded_blocks_match_whole(tmp_path):
    """config 5 says "sharded across 8 GPUs": contiguous blocks balanced by bases (bench.py / th_run_file_multi use the same
    rule) give the same records as the whole set in one batch"""
    import bench
    n = 80
    seqs, ad = U.config5_spec(tmp_path, n)
    t = TH.TideHunter(device=0, adapters=ad)
    whole = t.process(seqs)
    cuts = bench.partition_by_bases([len(s) for s in seqs], 8)
    assert cuts[0] == 0 and cuts[-1] == n and all(a < b for a, b in zip(cuts, cuts[1:]))
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        parts += t.process(seqs[a:b])
    assert parts == whole
    t.close()


@pytest.mark.parametrize("period,copies,err,seed,full,per_core", [
    (1000, 10, 0.10, 1, 2048, 128),    # config 3
    (200, 20, 0.10, 2, 2048, 128),     # config 2
    (5000, 4, 0.15, 3, 256, 16),       # config 4 (a job costs ~5-10 s and ~0.5 GB on the CPU)
])
def test_gpu_poa_equals_independent_exact_poa(period, copies, err, seed, full, per_core):
    """HIP POA kernel vs full-matrix Kahn-order POA on every job: the int16 / packed rows and the incremental node order of the
    kernel never change a consensus, and the adaptive band (part of the algorithm) does so on at most 0.5 % of the jobs"""
    n = _n(full, per_core)
    lib = TH.load_library()
    seqs = [TH.synth_read(lib, seed + 100, i, period, copies, err) for i in range(n)]
    t = TH.TideHunter(device=0)
    got = t.process(seqs)
    t.close()
    para = O.default_para()
    jobs = []
    for s, rr in zip(seqs, got):
        for r in rr:
            u = O.units_of(s, r["sub_pos"])
            if len(u) > 2:
                jobs.append((u, r["seq"]))
    assert len(jobs) >= n * 3 // 4

    def one(job):
        u, rec = job
        d, _ = O.poa_exact(u, para, 0)
        return d == rec

    with ThreadPoolExecutor(min(CORES, 64 if period >= 5000 else 256)) as ex:
        ok = list(ex.map(one, jobs))
    # A job may differ from the exact POA only through the adaptive band, which is part of the published algorithm (first seen
    # on the GPU box: 1 of 2053 config-2 jobs, a 3-unit run of period ~300): then the kernel must equal the banded restatement,
    # and the restatement WITHOUT its band must equal the exact POA (i.e. the node order is never the cause).
    lib = O.load()
    band_only = 0
    for (u, rec), good in zip(jobs, ok):
        if good:
            continue
        b, _ = O.poa_banded(u, para)
        try:
            lib.tho_poa_set_opt(O.POA_OPT_BAND, 0)
            ub, _ = O.poa_banded(u, para)
        finally:
            lib.tho_poa_set_opt(O.POA_OPT_BAND, 1)
        d, _ = O.poa_exact(u, para, 0)
        assert rec == b, "HIP kernel differs from the banded restatement"
        assert ub == d, "the node order changed a consensus (unbanded restatement != exact POA)"
        band_only += 1
    assert band_only * 200 <= len(jobs), "%d of %d POA jobs are changed by the adaptive band" % (band_only, len(jobs))


def test_gpu_structurizer_flag_changes_no_result(tmp_path):
    """the product is compiled with `-mllvm -structurizecfg-skip-uniform-regions=true` (csrc/Makefile), an LLVM-internal switch that upstream keeps off --
    every wave-uniform region of every kernel is left unstructured, and correctness rests on divergence analysis calling only truly uniform regions uniform.  The same sources
    built WITHOUT the flag (`make -C tidehunter_amd/csrc nostruct` -> tests/cpu_emul/libth_nostruct.so; built here if the snapshot does not carry it) must give the same bytes:
    ~2000 reads over four shapes -- the headline shape, short periods, long units on the relative-int16 POA and the six-register alignment variant, and config 5's mixed periods
    with adapters on both strands -- through th_format_output (the text mini_tandem_output prints, src/main.c:214-271), md5 per shape."""
    import hashlib
    ns = os.path.join(U.ROOT, "tests", "cpu_emul", "libth_nostruct.so")
    if not os.path.exists(ns) or os.path.getmtime(ns) < os.path.getmtime(os.path.join(U.ROOT, "tidehunter_amd", "libtidehunter_amd.so")) - 3600:
        subprocess.run(["make", "-s", "-C", os.path.join(U.ROOT, "tidehunter_amd", "csrc"), "nostruct"], check=True, timeout=900)
    lib = TH.load_library()
    c5, ad = U.config5_spec(tmp_path, 500)
    shapes = {"c3": ([TH.synth_read(lib, 61, i, 1000, 10, 0.10) for i in range(768)], None),
              "c2": ([TH.synth_read(lib, 62, i, 200, 20, 0.10) for i in range(512)], None),
              "c4": ([TH.synth_read(lib, 63, i, 5000, 4, 0.15) for i in range(96)] + [TH.synth_read(lib, 64, i, 2500, 5, 0.12) for i in range(96)], None),
              "c5": (c5, ad)}
    md5 = {}
    for which, path in (("flag_on", None), ("flag_off", ns)):
        for name, (reads, adapters) in shapes.items():
            t = TH.TideHunter(device=0, lib_path=path, adapters=adapters)
            t.process(reads)
            text = t.format_output([b"r%d" % i for i in range(len(reads))])
            st = t.stats()
            t.close()
            assert st["jobs_without_consensus"] == 0 and len(text) > 100 * len(reads) // 2
            md5[(which, name)] = hashlib.md5(text).hexdigest()
    print("structurizer flag on / off md5:", {n: (md5[("flag_on", n)], md5[("flag_off", n)]) for n in shapes})
    for name in shapes:
        assert md5[("flag_on", name)] == md5[("flag_off", name)], "%s: the two builds print different bytes" % name


def _bench_ranks(nproc, port, *args):
    """bench.py itself with the REAL library, `nproc` gloo ranks sharing device 0 (TH_BENCH_BACKEND=gloo TH_BENCH_DEVICE=0): one JSON line"""
    import json
    import sys
    env = dict(os.environ, TH_BENCH_BACKEND="gloo", TH_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TH_BENCH_LIB"):
        env.pop(k, None)
    common = ["--config", "c3", "--steps", "1", "--warmup", "1", "--digest", "--cpu-sample", "0", "--e2e", "0", "--streamed", "0", "--others", "", *args]
    bench = os.path.join(U.ROOT, "bench.py")
    if nproc == 1:
        cmd = [sys.executable, bench, "--gpus", "1", *common]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port), bench, "--gpus", str(nproc), *common]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


def test_gpu_bench_two_ranks_on_one_device_match_one_rank():
    """the multi-rank path of bench.py had only ever run on the CPU-emulation library.  Here the REAL library runs multi-process -- two ranks under
    torch.distributed.run, each with its own handle on device 0, gloo for the barrier / sums / ordered gather (no second GPU on this box: RCCL is what the driver's SCALE run
    adds) -- on 2 x 768 reads of the headline shape, static blocks by cumulative bases and blocks stolen through the shared ticket counter: the record digest and the md5 of
    the gathered text equal the one-rank run's over the same global list (src/main.c:273-291,402-425: the reference's chunk loop, one process)."""
    one = _bench_ranks(1, 29561, "--batch", "1536")
    two = _bench_ranks(2, 29562, "--batch", "768")
    steal = _bench_ranks(2, 29563, "--batch", "768", "--steal", "3")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and steal["n_gpus"] == 2
    assert one["records_digest_reads"] == two["records_digest_reads"] == steal["records_digest_reads"] == 1536
    assert one["records_digest"] == two["records_digest"] == steal["records_digest"]
    assert two["result_gather"]["backend"] == "gloo" and len(two["result_gather"]["bytes_by_rank"]) == 2 and all(b > 0 for b in two["result_gather"]["bytes_by_rank"])
    assert two["result_gather"]["output_md5"] == one["result_gather"]["output_md5"] and two["result_gather"]["bytes"] == one["result_gather"]["bytes"] > 1000000
    assert one["n_jobs_without_consensus"] == two["n_jobs_without_consensus"] == steal["n_jobs_without_consensus"] == 0
