#!/usr/bin/env python3
"""bench.py -- reads/s of the TideHunter hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (encode -> self-hits -> chain -> partition -> POA ->
unit/consensus alignments -> host record arithmetic) over one batch of synthetic reads.  Default
workload: BASELINE.json configs[2], the headline (period 1000 x 10 copies, 10 % error, ~10 kb reads,
100 000 reads per step per GPU = the config's whole read set); --config c2 / c4 / c5 select the other shapes of SURVEY.md 8(d).
The config's read set is generated ONCE per rank (deterministic generator, ~1 GB at the headline shape) and every step processes
the same batch drawn from it; the batch is uploaded to HBM before the timed region (inputs resident), and the K timed passes run back to back in ONE region
(th_batch_run_steps: the host-side record arithmetic of pass k overlaps the kernels of pass k+1, the last pass's ends inside the region).

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run (started from here as a child process when
WORLD_SIZE is not set).  The step's global read list (N x batch reads, the set repeated) is cut into N contiguous blocks balanced
by CUMULATIVE BASES; the data path has no collective -- RCCL carries the barrier and the max / sum of the times and
counts.  --steal K cuts the list into N*K blocks that ranks take from a shared ticket counter (th_ticket_*, POSIX shared memory;
the reference's THREAD_READ_I, src/main.c:277-281): then uploads happen inside the timed region and the line says so.

Prints ONE JSON line (rank 0) with `roofline` (the kernel with the largest device time, HIP-event timed on the
kernel's own stream inside the library), `cpu_baseline` (the CPU oracle -- a scalar C port of the reference's algorithm,
NOT the reference binary -- on the host cores, N=1 only), `value_upload_inclusive` (the same steps through th_process_batches(): H2D of
batch i+1 and the record arithmetic of batch i-1 overlap the kernels of batch i; inputs NOT resident), `value_e2e` (the command
line on 20 x the read set, first byte parsed to last byte written, every run listed, cold and steady rates apart) and
`other_shapes` (configs 2 / 4 / 5, three steps each, outside the headline's timed region) -- the last three at N=1 only.
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured copy ceiling quoted in SURVEY.md 8(d)

# name -> (period, copies, err, seed, reads in the config's set, default reads per step per GPU)
CONFIGS = {
    "c2": dict(period=200, copies=20, err=0.10, seed=2, set_size=10000, batch=65536,
               label="C2: period 200 x 20 copies, 10 percent error, ~4 kb reads (chain-dominated)"),
    "c3": dict(period=1000, copies=10, err=0.10, seed=1, set_size=100000, batch=100000,
               label="C3 headline: synthetic tandem-repeat reads, period 1000 x 10 copies, 10 percent error (sub/ins/del), ~10 kb"),
    "c4": dict(period=5000, copies=4, err=0.15, seed=3, set_size=20000, batch=20000,
               label="C4: period 5000 x 4 copies, 15 percent error, ~20 kb reads (long period)"),
    "c5": dict(period=None, copies=None, err=0.10, seed=5, set_size=100000, batch=100000,
               label="C5: periods {200,500,1000,2000,5000} by read index, copies max(3, 10000 // period), 10 percent error, both strands, "
                     "5'/3' adapters (-5/-3: full-length path + adapter search)"),
    # ONE long-read point, OUTSIDE the contract (not a BASELINE.json config): rolling-circle / satellite-like reads of 100 kb.  Reads of 65 536 bases and more
    # leave the packed chain-DP records and the 32-bit LDS sort words (64-bit words, HBM bitonic sort beyond 16 384 k-mers): the one regime where a fallback path could dominate
    "c_long": dict(period=2000, copies=50, err=0.10, seed=7, set_size=2000, batch=2000,
                   label="c_long -- NOT a BASELINE.json config: 2000 synthetic reads of ~100 kb, period 2000 x 50 copies, 10 percent error (long-read fallback paths)"),
    # rehearsal shape for the CPU-emulation tests of the multi-rank path (never a bench line)
    "tiny": dict(period=60, copies=6, err=0.08, seed=3, set_size=64, batch=4, label="tiny rehearsal shape (tests only)"),
}


def csrc_hash():
    """identifies the kernel sources a traffic profile was taken with: md5 over the contents of tidehunter_amd/csrc/*.{hip,h} (name-sorted).  tools/hbm_traffic.sh
    stores it in the profile; the bench line marks `roofline.traffic_stale` when the running build's differs."""
    h = hashlib.md5()
    d = os.path.join(ROOT, "tidehunter_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    # ... and the compiler flags (one of them is an LLVM-internal switch that changes the code of every kernel; a profile taken under other flags is stale too)
    try:
        h.update(subprocess.run(["make", "-s", "--no-print-directory", "-C", d, "print-hipflags"], capture_output=True, timeout=30).stdout.strip())
    except (OSError, subprocess.SubprocessError):
        h.update(b"hipflags unknown")
    return h.hexdigest()[:16]


# SURVEY.md 8(d), "secondary ceiling": these kernels are integer DP bound by VECTOR INSTRUCTION ISSUE, not by bytes.  A SIMD issues one vector wave-instruction of the class
# the DP bodies are made of (max, compare, v_pk_*, DPP, VOP3, lane ops) every 4.2 cycles -- measured on this chip, tools/valu_rate.hip -> profiles/r05_valu_rate.txt --
# so 1024 SIMDs x 2.39 GHz / 4.2 is the most such instructions per second the device retires.
ISSUE_SIMDS, ISSUE_CLOCK_HZ, ISSUE_CYCLES = 1024, 2.39e9, 4.2
SECONDARY_KERNELS = (("th_k_poa", "th_k_poa<"), ("th_k_align", "th_k_align<"), ("th_k_chain_dp", "th_k_chain_dp<"))


def newest_profile(suffix):
    """(dict, file name) of the newest profiles/rNN_<suffix> (round number descending), or (None, None)"""
    pd = os.path.join(ROOT, "profiles")
    try:
        names = sorted((f for f in os.listdir(pd) if re.match(r"r\d\d_" + re.escape(suffix) + "$", f)), reverse=True)
    except OSError:
        names = []
    for f in names:
        try:
            return json.load(open(os.path.join(pd, f))), f
        except (OSError, ValueError):
            continue
    return None, None


def secondary_roofline(kms, launches, reads_per_launch, config, my_hash, steal):
    """`roofline.secondary`: per big kernel the vector / scalar wave-instructions of one launch (quoted from the newest profiles/rNN_pmc_instr.json, a rocprofv3 --pmc pass of this
    same command; marked stale when the kernel sources or flags have changed since), the issue ceiling they are priced against, issue_frac = vector instructions / (ceiling x the
    launch time measured LIVE in this run), the DP cells of one launch (profiles/rNN_dp_cells.json: device counters of the profiling build -- a property of the workload), cells / s
    and vector instructions per cell."""
    ceiling = ISSUE_SIMDS * ISSUE_CLOCK_HZ / ISSUE_CYCLES
    pmc, pmc_fn = newest_profile("pmc_instr.json")
    cells, cells_fn = newest_profile("dp_cells.json")
    ok_pmc = bool(pmc) and pmc.get("reads_per_launch", 100000) == reads_per_launch and pmc.get("config", "c3") == config and steal == 0
    ok_cells = bool(cells) and cells.get("reads_per_launch") == reads_per_launch and cells.get("config") == config and steal == 0
    out = {"bound": "vector_issue", "unit": "wave-instructions/s", "peak": ceiling,
           "peak_is": "%d SIMDs x %.2f GHz / %.1f cycles per vector wave-instruction of the DP class (profiles/r05_valu_rate.txt)" % (ISSUE_SIMDS, ISSUE_CLOCK_HZ / 1e9, ISSUE_CYCLES),
           "instr_source": ("profiles/" + pmc_fn) if ok_pmc else None, "instr_stale": (pmc.get("csrc_hash") != my_hash) if ok_pmc else None,
           "cells_source": ("profiles/" + cells_fn) if ok_cells else None, "kernels": {}}
    tot_cells, tot_s = 0.0, 0.0
    for grp, pat in SECONDARY_KERNELS:
        launch_s = kms.get(grp, 0.0) / 1000.0 / launches
        k = {"launch_ms": launch_s * 1000.0}
        if ok_pmc:
            rows = [v for n, v in pmc["kernels"].items() if pat in n]
            if rows:
                valu, salu = sum(r["valu_wave_instr"] / max(r["launches"], 1) for r in rows), sum(r["salu_wave_instr"] / max(r["launches"], 1) for r in rows)
                k.update({"valu_wave_instr": valu, "salu_wave_instr": salu, "achieved": valu / launch_s if launch_s > 0 else None,
                          "issue_frac": valu / launch_s / ceiling if launch_s > 0 else None,
                          "scalar_issue_frac": salu * 4.25 / ISSUE_CYCLES / launch_s / ceiling if launch_s > 0 else None})
        if ok_cells and grp in cells["kernels"]:
            c = float(cells["kernels"][grp]["dp_cells_per_launch"])
            k.update({"dp_cells": c, "gcells_per_s": c / launch_s / 1e9 if launch_s > 0 else None, "cell": cells["kernels"][grp].get("cell")})
            if k.get("valu_wave_instr"):
                k["valu_lane_instr_per_cell"] = k["valu_wave_instr"] * 64.0 / c          # a wave-instruction is 64 lane-instructions
            tot_cells += c; tot_s += launch_s
        out["kernels"][grp] = k
    out["total_gcells_per_s"] = tot_cells / tot_s / 1e9 if tot_s > 0 else None
    return out


def effective_cpus():
    """CPUs this process can actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 logical CPUs
    to nproc but schedule the container on a 16-CPU quota; threads beyond the quota only add throttling)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:                                                                   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def partition_by_bases(lens, parts):
    """cuts[0..parts]: contiguous blocks of the read list balanced by cumulative bases (cost is ~ bases); block b
    is reads cuts[b]:cuts[b+1].  Every block is non-empty when there are at least `parts` reads."""
    import numpy as np
    n = len(lens)
    if parts <= 1 or n == 0:
        return [0, n]
    cum = np.cumsum(np.asarray(lens, dtype=np.int64))
    total = int(cum[-1])
    cuts = [0]
    for b in range(1, parts):
        c = int(np.searchsorted(cum, total * b / parts, side="left")) + 1      # first prefix holding at least b/parts of the bases
        c = max(c, cuts[-1] + 1) if n >= parts else max(c, cuts[-1])
        cuts.append(min(c, n - (parts - b) if n >= parts else n))
    cuts.append(n)
    return cuts


def cpu_baseline(read_at, n_reads, cfg):
    """CPU oracle (oracle/tho_oracle, kind 'port') on the first n_reads of the same read set, with one thread and with one
    thread per core; reports the per-core cost both ways so that the thread scaling is visible."""
    cores = effective_cpus()
    exe = os.path.join(ROOT, "oracle", "tho_oracle")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "tho_oracle"], check=True)
    path = "/tmp/th_bench_cpu_sample_%d.fa" % os.getpid()
    n1 = max(8, min(48, n_reads // 16))
    with open(path, "wb") as f:
        for i in range(n_reads):
            f.write(b">syn_%d\n" % i + read_at(i) + b"\n")
    with open(path + ".1", "wb") as f:
        for i in range(n1):
            f.write(b">syn_%d\n" % i + read_at(i) + b"\n")
    extra = cfg.get("cli_opts", [])

    def run(threads, fn):
        t0 = time.time()
        err = subprocess.run([exe, "-t", str(threads), *extra, "-o", "/dev/null", fn], check=True, capture_output=True, text=True).stderr
        m = re.search(r"threads: (\d+); ([0-9.]+) reads/s; ([0-9.]+) ms per read per core", err)
        return float(m.group(2)), float(m.group(3)), time.time() - t0

    r1, ms1, _ = run(1, path + ".1")
    rn, msn, dt = run(cores, path)
    os.unlink(path); os.unlink(path + ".1")
    return {"value": rn, "unit": "reads/s", "cores": cores, "kind": "port", "logical_cpus_visible": os.cpu_count(),
            "cores_note": "cores = CPUs this container may use (affinity mask capped by the cgroup CPU quota), one oracle thread each",
            "single_thread": {"reads_per_s": r1, "ms_per_read_per_core": ms1, "reads": n1},
            "all_threads": {"reads_per_s": rn, "ms_per_read_per_core_cpu_time": msn, "ms_per_read_x_threads_wall": 1000.0 * cores / rn, "threads": cores},
            "thread_scaling_efficiency": rn / (r1 * cores) if r1 > 0 else None,
            "sample": "%d reads of the same set, oracle/tho_oracle -t %d, %.1f s (and %d reads at -t 1)" % (n_reads, cores, dt, n1),
            "note": "scalar C restatement of the reference's algorithm (oracle/; no SIMD, its POA is a restatement because abPOA is absent), "
                    "NOT the reference binary: a GPU/port ratio says nothing about the reference.  Reference-derived context (SURVEY.md "
                    "section 6, measured on the reference's own code, 2.1 GHz Xeon): stages 1-3 + all ksw2 work cost 20 ms per read per "
                    "core at this shape WITHOUT abPOA, so an ideal %d-core host runs at most %.0f reads/s before any abPOA time." % (cores, cores * 50.0)}


def cli_e2e(set_reads, copies, cfg, batch_opt, n_runs=2, pause_s=6.0):
    """value_e2e: the command line on `copies` x the read set (2 000 000 reads at the headline shape), wall-clock from the first byte
    parsed to the last byte written (th_run_file's own clock; process start and HIP initialisation are reported separately as
    `wall_with_startup_s`).  Every run is listed: the first is COLD (first GPU process of this bench: HIP initialisation, device
    allocation, page cache), `steady_reads_per_s` is the rate after the first full-size batch was complete (start-up excluded)."""
    exe = os.path.join(ROOT, "tidehunter_amd", "TideHunter-amd")
    set_bytes = sum(len(s) for s in set_reads)
    # input + warm-up input + output need ~1.25 x the FASTA: the first directory with that much room (tmpfs preferred), fewer copies if none has
    d = None
    while d is None and copies >= 1:
        for cand in ("/dev/shm", "/tmp"):
            try:
                st = os.statvfs(cand)
                if os.path.isdir(cand) and st.f_bavail * st.f_frsize > 1.25 * (copies + 1) * (set_bytes + 16 * len(set_reads)):
                    d = cand
                    break
            except OSError:
                pass
        if d is None:
            copies //= 2
    if d is None:
        return {"error": "no room for the end-to-end input in /dev/shm or /tmp"}
    path, outp = os.path.join(d, "th_bench_e2e_%d.fa" % os.getpid()), os.path.join(d, "th_bench_e2e_%d.out" % os.getpid())
    n_reads, nbytes = copies * len(set_reads), copies * set_bytes
    runs = []
    try:
        with open(path, "wb") as f:
            for c in range(copies):
                f.write(b"".join(b">syn_%d\n" % (c * len(set_reads) + i) + s + b"\n" for i, s in enumerate(set_reads)))
        # the first GPU process on a machine pays for things that are not this program's (driver and runtime start-up -- seen: 4 s until the
        # first handle is ready -- and the first large device allocation on untouched HBM -- seen: 3.5 s): a run over one small and one full
        # batch takes them, listed apart with its own rate and not part of `value`
        wpath, n_warm = path + ".warm", min(len(set_reads), 2 * batch_opt + batch_opt // 4)       # (both pipelines of the device see a full batch)
        with open(wpath, "wb") as f:
            f.write(b"".join(b">w_%d\n" % i + s + b"\n" for i, s in enumerate(set_reads[:n_warm])))
        t0 = time.time()
        subprocess.run([exe, *cfg.get("cli_opts", []), "-b", str(batch_opt), "-o", outp, wpath], check=True, capture_output=True)
        machine_warmup = {"reads": n_warm, "wall_s": time.time() - t0, "reads_per_s_with_process_startup": n_warm / (time.time() - t0)}
        os.unlink(wpath)
        for run_i in range(n_runs + 1):
            # the driver is still clearing the HBM the previous process released (70-140 GB) for a few seconds after its exit, and a new
            # process's first allocation waits for that -- 1.5-4.5 s seen back to back (profiles/r03_e2e_pipelines.txt); it is the previous
            # run's cost, not this one's, so each of the n_runs runs behind `value` starts after a pause.  One more run follows WITHOUT a
            # pause and is listed apart (`run_without_pause`): what a caller sees who starts the next file the moment the last one ended.
            pause = pause_s if run_i < n_runs else 0.0
            time.sleep(pause)
            t0 = time.time()
            err = subprocess.run([exe, *cfg.get("cli_opts", []), "-b", str(batch_opt), "-o", outp, path], check=True, capture_output=True, text=True,
                                 env=dict(os.environ, TH_TRACE="1")).stderr
            wall = time.time() - t0
            m = re.search(r"\[th_run_file\] (\d+) reads on \d+ device pipeline\(s\): wall ([0-9.]+) s; busy: reader ([0-9.]+) s, upload ([0-9.]+) s, device runs ([0-9.]+) s, "
                          r"record arithmetic ([0-9.]+) s, writer ([0-9.]+) s; steady ([0-9.]+) reads/s over the last ([0-9.]+) s", err)
            inner = float(m.group(2))
            assert int(m.group(1)) == n_reads, (m.group(1), n_reads)
            mc = re.search(r"\[th_run_file\] host CPU ([0-9.]+) s \(([0-9.]+) per 10\^6 reads; (\d+) usable CPUs\): reader ([0-9.]+), inflate ([0-9.]+), upload threads ([0-9.]+), "
                           r"run threads ([0-9.]+), record arithmetic \+ formatting ([0-9.]+), writer ([0-9.]+), runtime and the rest (-?[0-9.]+)", err)
            mp = re.search(r"\[main\] Real time: ([0-9.]+) sec; CPU: ([0-9.]+) sec", err)
            host_cpu = None
            if mc:
                host_cpu = {"cpu_s": float(mc.group(1)), "cpu_s_per_1M_reads": float(mc.group(2)), "usable_cpus": int(mc.group(3)),
                            "by_role_s": {"reader": float(mc.group(4)), "inflate": float(mc.group(5)), "upload_threads": float(mc.group(6)), "run_threads": float(mc.group(7)),
                                          "record_arithmetic_and_formatting": float(mc.group(8)), "writer": float(mc.group(9)), "runtime_and_rest": float(mc.group(10))},
                            "process_cpu_s_incl_startup": float(mp.group(2)) if mp else None, "cores_busy_mean": float(mc.group(1)) / inner}
            n_fail = sum(int(x) for x in re.findall(r"warning: (\d+) consensus job\(s\) could not be aligned", err))
            md5 = hashlib.md5()
            with open(outp, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 24), b""):
                    md5.update(chunk)
            runs.append({"reads_per_s": n_reads / inner, "seconds_first_byte_to_last_byte": inner, "wall_with_startup_s": wall,
                         "steady_reads_per_s": float(m.group(8)), "steady_window_s": float(m.group(9)), "startup_s": inner - float(m.group(9)),
                         "mbases_per_s": nbytes / inner / 1e6, "reader_busy_s": float(m.group(3)), "upload_busy_s": float(m.group(4)), "device_busy_s": float(m.group(5)),
                         "collect_busy_s": float(m.group(6)), "writer_busy_s": float(m.group(7)), "output_md5": md5.hexdigest(), "host_cpu": host_cpu, "n_jobs_without_consensus": n_fail,
                         "paused_before_s": pause})
        nopause = runs.pop()
        vals = [r["reads_per_s"] for r in runs]
        cpus = [r["host_cpu"]["cpu_s_per_1M_reads"] for r in runs if r.get("host_cpu")]
        return {"value": sum(vals) / len(vals), "unit": "reads/s", "value_is": "mean of the whole-run rates of all runs (first byte parsed to last byte written), no run dropped",
                "host_cpu_s_per_1M_reads": sum(cpus) / len(cpus) if cpus else None,
                "host_cpu_note": "CPU seconds of the whole process between the first byte parsed and the last byte written (getrusage), per million reads, mean of the runs; "
                                 "by role in runs[].host_cpu (thread CPU clocks; the remainder is the HIP runtime's own threads).  The seam this replaces keeps n_thread workers "
                                 "and one writer busy (src/main.c:273-291,214-271); round 3 spent 29.9 here, 16 of them in spinning stream synchronisations",
                "run_without_pause": nopause, "n_jobs_without_consensus": sum(r["n_jobs_without_consensus"] for r in runs) + nopause["n_jobs_without_consensus"],
                "reads": n_reads, "runs_reads_per_s": vals, "cold_reads_per_s": vals[0], "steady_reads_per_s": [r["steady_reads_per_s"] for r in runs],
                "runs": runs, "batch_reads": batch_opt, "machine_warmup_run": machine_warmup, "pause_before_each_run_s": pause_s,
                "what": "TideHunter-amd (reader -> H2D -> kernels -> record arithmetic -> writer) on a FASTA of %d x the read set in %s" % (copies, d)}
    finally:
        for p in (path, outp, path + ".warm"):
            if os.path.exists(p):
                os.unlink(p)


def other_shapes(names, steps=3):
    """configs 2 / 4 / 5 as three-step child runs of this script (outside the headline's timed region; N=1)"""
    out = {}
    for c in names:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", c, "--steps", str(steps), "--warmup", "1", "--cpu-sample", "0", "--e2e", "0",
                                "--others", "", "--streamed", "0"], capture_output=True, text=True, timeout=900)
            j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            out[c] = {"reads_per_s": j["value"], "mbases_per_s": j["mbases_per_s"], "ms_per_step": j["ms_per_step"], "reads_per_step": j["config"]["reads_per_step_per_gpu"],
                      "kernel_ms_per_step": j["roofline"]["kernel_ms_per_step"], "workload": j["config"]["workload"], "n_jobs_without_consensus": j.get("n_jobs_without_consensus")}
        except Exception as e:                                          # a shape that fails is reported, not hidden
            out[c] = {"error": repr(e)[:300]}
    return out


def spawn_ranks(a):
    """--gpus N without a launcher: start torch.distributed.run as a child BEFORE anything touches the GPU and pass its exit code on"""
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)   # two: device buffers reach their steady size (a batch that grows one is first-touched inside a kernel)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TH_BENCH_BATCH", "0")), help="reads per step per GPU (0: the config's default)")
    ap.add_argument("--steal", type=int, default=0, help="blocks per rank per step taken from a shared ticket counter (0: static partition by bases)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1: auto, 0: skip)")
    ap.add_argument("--e2e", type=int, default=-1, help="copies of the read set in the end-to-end command-line run (-1: 20 for the headline workload, 0: skip)")
    ap.add_argument("--streamed", type=int, default=-1, help="also time the steps through th_process_batches(), H2D inside (-1: at N=1, 0: skip)")
    ap.add_argument("--others", default=None, help="comma-separated configs for `other_shapes` (default: c2,c4,c5,c_long with the headline workload at N=1; '' to skip)")
    ap.add_argument("--digest", action="store_true", help="add an md5 over all records of the timed steps in read order (tests)")
    ap.add_argument("--quick", action="store_true", help="the timed steps only: no CPU baseline, end-to-end run, streamed run or other shapes (profiler passes)")
    a = ap.parse_args()
    if a.quick:
        a.cpu_sample, a.e2e, a.streamed, a.others = 0, 0, 0, ""

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                 "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...`" % (a.gpus, world, a.gpus, a.gpus))
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    # The contract is ONE JSON line on stdout.  Libraries under us write there too (torch's gloo backend announces its peers on
    # stdout, the ROCm runtime complains about a missing amdgpu.ids): from here on file descriptor 1 is stderr, and the result
    # line goes to the saved original stdout at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    cfg = dict(CONFIGS[a.config])
    B = a.batch if a.batch > 0 else cfg["batch"]
    # host threads of the record arithmetic: this rank's share of the CPUs the container may use (8 ranks on a 16-CPU quota get 2 each)
    os.environ.setdefault("TH_HOST_THREADS", str(max(1, min(16, effective_cpus() // world))))

    import torch
    dist = None
    # TH_BENCH_BACKEND / TH_BENCH_DEVICE / TH_BENCH_LIB rehearse the multi-rank path without N GPUs (gloo; every rank on one
    # device, or the CPU-emulation build of the kernels in tests/): tests/test_bench_multirank.py
    backend = os.environ.get("TH_BENCH_BACKEND", "nccl")
    on_gpu = backend == "nccl" or "TH_BENCH_LIB" not in os.environ
    if "TH_BENCH_DEVICE" in os.environ:
        local = int(os.environ["TH_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    import tidehunter_amd as TH
    lib = TH.load_library(os.environ.get("TH_BENCH_LIB"))
    adapters, kw = None, {}
    if a.config == "c5":
        adapters = TH.synth_adapters(lib)
        d = "/tmp/th_bench_c5_%d" % os.getpid()
        os.makedirs(d, exist_ok=True)
        fns = (os.path.join(d, "5p.fa"), os.path.join(d, "3p.fa"))
        open(fns[0], "wb").write(b">5p\n" + adapters[0] + b"\n"); open(fns[1], "wb").write(b">3p\n" + adapters[1] + b"\n")
        kw["adapters"] = fns
        cfg["cli_opts"] = ["-5", fns[0], "-3", fns[1]]

    G = B * world                                   # reads of one step's global list (weak scaling: B per GPU)
    n_blocks = world * max(1, a.steal)
    # the config's read set, generated once per rank and shared by every step (<= 1 GB at the headline shape): read i of a step's
    # global list is read i mod |set|
    t_gen = time.time()
    n_set = min(cfg["set_size"], G)
    if a.config == "c5":
        set_reads = [TH.synth_read_c5(lib, cfg["seed"], i, adapters, cfg["err"]) for i in range(n_set)]
    else:
        set_reads = [TH.synth_read(lib, cfg["seed"], i, cfg["period"], cfg["copies"], cfg["err"]) for i in range(n_set)]
    t_gen = time.time() - t_gen

    def read_at(i):
        return set_reads[i % n_set]

    # value_e2e runs FIRST, in child processes, before this process touches the GPU: its first run is then the first GPU process of
    # the bench (cold), and the HBM it releases is not being scrubbed under this process's own allocations during the timed steps
    e2e = None
    if rank == 0 and world == 1 and a.e2e != 0 and not (a.e2e < 0 and a.config != "c3") and "TH_BENCH_LIB" not in os.environ:      # (by default only for the headline workload)
        try:
            e2e = cli_e2e(set_reads, a.e2e if a.e2e > 0 else 20, cfg, 65536)      # (65536 = the command line's default batch; 20 x the set = 2 000 000 reads: start-up is ~0.2 s of each run)
        except Exception as ex:                       # reported, never fatal for the bench line
            e2e = {"error": repr(ex)[:400]}
    th = TH.TideHunter(device=local, lib_path=os.environ.get("TH_BENCH_LIB"), **kw)

    def sync_all():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    # block boundaries of a step's global list [0, G), balanced by cumulative bases: every rank holds the whole set, so every rank
    # computes the same cuts (no exchange); block k is reads cuts[k] .. cuts[k+1]
    set_lens = [len(s) for s in set_reads]
    cuts = partition_by_bases([set_lens[i % n_set] for i in range(G)], n_blocks) if n_blocks > 1 else [0, G]

    def block_reads(k):
        return [read_at(i) for i in range(cuts[k], cuts[k + 1])]            # (references into the set, not copies)

    n_batches = a.warmup + a.steps
    ticket = None
    if a.steal > 0:
        # one segment per launch: the name carries the launcher's port AND rank 0's pid, and the creator starts from a fresh
        # segment (a killed run can leave an old one behind with its counters advanced)
        nonce = torch.tensor([os.getpid()], dtype=torch.int64)
        if dist is not None:
            nonce = nonce.cuda() if backend == "nccl" else nonce
            dist.broadcast(nonce, 0)
        name = ("/th_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), int(nonce.item()))).encode()
        if rank == 0:
            ticket = lib.th_ticket_open(name, 1)
        if dist is not None:
            dist.barrier()
        if rank != 0:
            ticket = lib.th_ticket_open(name, 0)
        assert ticket, "th_ticket_open failed"
        if a.steps + a.warmup > 512:
            sys.exit("bench.py: --steal supports at most 512 steps")
    my_batch = None if a.steal > 0 else block_reads(rank)

    for b in range(a.warmup):
        if a.steal > 0:
            while True:
                k = lib.th_ticket_next(ticket, b)
                if k >= n_blocks:
                    break
                th.upload(block_reads(k)); th.run(); th.collect()
        else:
            th.upload(my_batch); th.run(); th.collect()
    t_total, n_cons, alg_bytes, kms, n_reads_mine, bases_mine, digests, blocks_mine, n_syncs = 0.0, 0, 0, {}, 0, 0, {}, 0, 0
    gather_info = None
    import resource

    def cpu_now():
        ru = resource.getrusage(resource.RUSAGE_SELF)
        return ru.ru_utime + ru.ru_stime
    cpu_timed = 0.0

    def add_timing():
        for kname, v in th.timing().items():
            kms[kname] = kms.get(kname, 0.0) + v

    def account(batch, recs, first_index):
        nonlocal n_cons, alg_bytes, n_reads_mine, bases_mine
        for j, (s, rr) in enumerate(zip(batch, recs)):
            alg_bytes += len(s)                                     # B_alg = L + sum(cons_len + 4*pos_n + 40)
            for r in rr:
                alg_bytes += r["cons_len"] + 4 * len(r["sub_pos"]) + 40
                n_cons += 1
            if a.digest:
                digests[first_index + j] = hashlib.md5(repr([(r["seq"], r["start"], r["end"], r["cons_len"], "%.6f" % r["copy_num"], "%.6f" % r["ave_match"],
                                                               r["full_length"], r["sub_pos"]) for r in rr]).encode()).hexdigest()
        n_reads_mine += len(batch)
        bases_mine += sum(len(s) for s in batch)

    if a.steal > 0:
        for b in range(a.warmup, n_batches):
            # dynamic: blocks from the shared ticket counter; the upload of a block cannot precede the timed region
            sync_all()
            t0 = time.perf_counter(); c0 = cpu_now()
            done = []
            while True:
                k = lib.th_ticket_next(ticket, b)
                if k >= n_blocks:
                    break
                batch = block_reads(k)
                th.upload(batch); th.run(); th.collect_native()
                add_timing()
                done.append((k, batch, th.records() if (a.digest or b == n_batches - 1) else None))
                blocks_mine += th.launches()
            if on_gpu:
                torch.cuda.synchronize()
            t_total += time.perf_counter() - t0; cpu_timed += cpu_now() - c0
            for k, batch, recs in done:
                if recs is not None:
                    account(batch, recs, cuts[k])
                else:
                    n_reads_mine += len(batch); bases_mine += sum(len(s) for s in batch)
    else:
        # static: this rank's block is made resident first (upload outside), then EXACTLY K passes are timed in one region:
        # th_batch_run_steps() runs all kernels + the small D2H of result records of pass k+1 while the host-side record arithmetic
        # (src/gen_cons.c:208-301, in C) of pass k fills the tandem_seq_t slots -- the last pass's arithmetic ends inside the region
        th.upload(my_batch)
        sync_all()
        t0 = time.perf_counter(); c0 = cpu_now()
        kstep = th.run_steps(a.steps)
        if on_gpu:
            torch.cuda.synchronize()
        t_total += time.perf_counter() - t0; cpu_timed += cpu_now() - c0
        n_syncs = th.syncs()
        for kname, v in kstep.items():
            kms[kname] = kms.get(kname, 0.0) + v
        blocks_mine += th.launches() * a.steps              # TH_SUBBATCH > 1: several concurrent parts, that many launches of every kernel
        account(my_batch, th.records(), cuts[rank])         # Python dict view for the byte accounting (every pass has the same records): not part of the path
        n_reads_mine += len(my_batch) * (a.steps - 1); bases_mine += sum(len(s) for s in my_batch) * (a.steps - 1)
        # ---- ordered result gather (outside the timed region): every rank formats the records of its block as the command line would print
        # them (th_format_output = mini_tandem_output, src/main.c:214-271, into memory) and the blocks go to rank 0 in rank order = input
        # order (src/main.c:217) -- torch.distributed gather of byte tensors: RCCL over xGMI at N > 1 on the GPUs, gloo in the rehearsals
        my_text = th.format_output([b"syn_%d" % gi for gi in range(cuts[rank], cuts[rank + 1])])
        sync_all()
        tg0 = time.perf_counter()
        if dist is not None:
            import numpy as np
            dev = "cuda" if backend == "nccl" else "cpu"
            sizes = torch.zeros(world, dtype=torch.int64, device=dev); sizes[rank] = len(my_text)
            dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
            sizes = [int(x) for x in sizes.cpu()]
            cap = max(max(sizes), 1)
            buf = torch.zeros(cap, dtype=torch.uint8)
            buf[: len(my_text)] = torch.from_numpy(np.frombuffer(my_text, dtype=np.uint8).copy()) if my_text else buf[:0]
            buf = buf.to(dev)
            parts = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
            dist.gather(buf, parts, dst=0)
            if on_gpu:
                torch.cuda.synchronize()
            gathered = b"".join(bytes(parts[r][: sizes[r]].cpu().numpy()) for r in range(world)) if rank == 0 else b""
        else:
            sizes, gathered = [len(my_text)], my_text
        gather_s = time.perf_counter() - tg0
        gather_info = {"bytes": sum(sizes), "bytes_by_rank": sizes, "seconds": gather_s, "backend": (backend if dist is not None else "none (one rank)"),
                       "order": "contiguous blocks in rank order = input order", "output_md5": hashlib.md5(gathered).hexdigest() if rank == 0 else None,
                       "what": "the FASTA text of the last step's records of every rank (th_format_output), gathered to rank 0 with torch.distributed.gather; outside the timed region"}
    sync_all()
    st = th.stats()          # cumulative over warm-up, timed steps (and nothing else yet): jobs the POA kernels could not serve must be 0
    # the byte accounting ran on the last step (all steps process the same reads): scale it to the timed steps
    if (a.steal == 0 or not a.digest) and a.steps > 1:
        n_cons *= a.steps; alg_bytes *= a.steps
    tot = torch.tensor([float(n_reads_mine), float(bases_mine), float(n_cons), float(alg_bytes), cpu_timed, float(st["jobs_without_consensus"]), float(st["poa_jobs_rerun"]), float(st["stage_reruns"]), float(st["jobs"])], dtype=torch.float64)
    cpu_by_rank = [cpu_timed]
    if dist is not None:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([t_total], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_total = float(t.item())
        tot = tot.to(dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot = tot.cpu()
        cpu_by_rank = [None] * world
        dist.all_gather_object(cpu_by_rank, cpu_timed)
        if a.digest:
            gathered = [None] * world
            dist.all_gather_object(gathered, digests)
            digests = {k: v for g in gathered for k, v in g.items()}
    reads, bases, n_cons_all = int(tot[0].item()), tot[1].item(), tot[2].item()
    assert reads == G * a.steps, (reads, G, a.steps)
    value = reads / t_total
    out = {
        "metric": "reads/sec on 10 kb period-1000x10 reads (TideHunter hot path)" if a.config == "c3" else "reads/sec (TideHunter hot path, %s)" % a.config,
        "value": value, "unit": "reads/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * t_total / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "dtype_note": "integer DP throughout: chain DP and POA fallback rows in int32, packed POA rows and banded alignments in int16 pairs (exact by range proofs)",
        "mbases_per_s": bases / t_total / 1e6,
        "config": {"workload": cfg["label"] + "; %d reads per step per GPU drawn from the %d-read set (seed %d), the same batch every step" % (B, cfg["set_size"], cfg["seed"]),
                   "name": a.config, "reads_per_step_per_gpu": B,
                   "sharding": ("contiguous read blocks per rank balanced by cumulative bases, no data-path collective" if a.steal == 0 else
                                "%d blocks per step (balanced by cumulative bases) taken by the ranks from a shared ticket counter; uploads inside the timed region" % n_blocks),
                   "inputs_resident_before_timing": a.steal == 0,
                   "value_note": "`value` is the device path with the step's reads already in HBM, as the bench contract defines it; the same steps fed from HOST memory are "
                                 "`value_upload_inclusive` (PCIe inside, N=1), and `value_e2e` is the command line from the first byte parsed to the last byte written"},
        "consensus_per_read": n_cons_all / float(reads),
        # recovery paths over warm-up + timed steps, all ranks: consensus jobs the POA kernels could not serve (reported WITHOUT a consensus: output that differs
        # from the reference's -- must be 0), POA jobs given back and rerun on a slower path, reruns of stages 2-3 because an arena or table was too small
        "n_jobs_without_consensus": int(tot[5].item()), "recovery_paths": {"poa_jobs_rerun": int(tot[6].item()), "stage_reruns": int(tot[7].item()), "consensus_jobs": int(tot[8].item())},
        "host_syncs_per_step": n_syncs, "read_set_generation_s": t_gen,
        # CPU seconds of each rank's process inside the timed region (getrusage; the record arithmetic threads and the runtime's own included), and
        # the same per million reads of the whole job: the host budget at N ranks is effective_cpus() / N cores each
        "host_cpu": {"cpu_s_per_1M_reads": float(tot[4].item()) * 1e6 / float(reads), "cpu_s_in_timed_region_by_rank": [round(float(x), 3) for x in cpu_by_rank],
                     "cores_busy_mean_per_rank": float(tot[4].item()) / world / t_total, "usable_cpus": effective_cpus(), "host_threads_per_rank": int(os.environ.get("TH_HOST_THREADS", "0"))},
    }
    if gather_info is not None:
        out["result_gather"] = gather_info
    if a.digest:
        h = hashlib.md5()
        for k in sorted(digests):
            h.update(digests[k].encode())
        out["records_digest"] = h.hexdigest()
        out["records_digest_reads"] = len(digests)
    if rank == 0:
        dom = max(kms, key=kms.get)                      # dominant kernel = largest device time (this rank's launches)
        launches = max(1, blocks_mine)
        launch_s = kms[dom] / 1000.0 / launches            # one launch of it per block (or per concurrent part of a block)
        per_launch_bytes = alg_bytes / float(launches)
        ach = per_launch_bytes / launch_s / 1e9 if launch_s > 0 else 0.0
        # traffic: HBM bytes of one launch of the dominant kernel from the PMC passes of this same command
        # (tools/round4_end.sh -> profiles/rNN_hbm_traffic.json); counters cannot be read from inside the process, so
        # the committed profile is quoted when it was taken at the same workload and batch size, else null
        traffic, traffic_lo, traffic_src, traffic_stale = None, None, None, None
        my_hash = csrc_hash()
        for pf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d\d_hbm_traffic.json$", f)), reverse=True)[:2]:
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", pf)))
                if prof.get("reads_per_launch") == B * a.steps // launches and prof.get("config", "c3") == a.config and a.steal == 0:
                    for kname, v in prof["kernels"].items():
                        if traffic is None and kname.replace("void ", "").split("<")[0].split("::")[-1] == dom:
                            traffic, traffic_lo = v["hbm_bytes_per_launch_corrected"], v.get("hbm_bytes_per_launch_uncorrected")
                            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % pf
                            # the profile names the kernel sources it was taken with: a kernel changed since then quotes stale bytes, and the line says so
                            traffic_stale = prof.get("csrc_hash") != my_hash
            except (OSError, ValueError, KeyError):
                pass
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "frac_of_measured_copy_peak": ach / HBM_COPY_GBS, "measured_copy_peak": HBM_COPY_GBS,
                           "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "csrc_hash": my_hash, "launch_ms": launch_s * 1000.0,
                           # what the HBM system really moves for this kernel: measured traffic / launch time, both bounds of the gfx950 FETCH correction
                           "achieved_measured": [traffic_lo / launch_s / 1e9, traffic / launch_s / 1e9] if (traffic and traffic_lo and launch_s > 0) else None,
                           "frac_measured": [traffic_lo / launch_s / 1e9 / HBM_PEAK_GBS, traffic / launch_s / 1e9 / HBM_PEAK_GBS] if (traffic and traffic_lo and launch_s > 0) else None,
                           "traffic_over_algorithmic": traffic / per_launch_bytes if (traffic and per_launch_bytes) else None,
                           "kernel_ms_per_launch": {k: v / launches for k, v in kms.items()},
                           "kernel_ms_per_step": {k: v / a.steps for k, v in kms.items()},
                           "algorithmic_bytes_per_launch": per_launch_bytes, "launches_per_step": launches / float(a.steps),
                           "note": "`achieved` / `frac` price the ALGORITHMIC bytes (the contract's definition); `achieved_measured` / `frac_measured` are the HBM bytes the counters saw "
                                   "(uncorrected, corrected) over the same launch time.  The path is bound by vector instruction issue: `secondary` prices that (SURVEY.md 8(d))"}
        out["roofline"]["secondary"] = secondary_roofline(kms, launches, B * a.steps // launches, a.config, my_hash, a.steal)
        # the same steps with the upload INSIDE: th_process_batches() pipelines H2D of batch i+1 and the record arithmetic of batch i-1
        # under the kernels of batch i (what th_run_file does with batches from the reader)
        if world == 1 and a.steal == 0 and a.streamed != 0:
            ks = max(a.steps, 8)                       # (fill and drain of the three-stage pipeline are ~0.15 s: eight batches keep them near 2 %)
            ctx = th.prepare_batches([my_batch] * ks)          # (argument arrays and result slots: what a C caller owns before the call)
            t0 = time.perf_counter(); c0 = cpu_now()
            th.run_prepared(ctx)
            if on_gpu:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0; c1 = cpu_now()
            th.free_prepared(ctx)
            out["value_upload_inclusive"] = {"value": ks * len(my_batch) / dt, "unit": "reads/s", "steps": ks, "ms_per_step": 1000.0 * dt / ks, "inputs_resident_before_timing": False,
                                             "ratio_to_value": ks * len(my_batch) / dt / value, "host_cpu_s_per_1M_reads": (c1 - c0) * 1e6 / (ks * len(my_batch)),
                                             "what": "th_process_batches(): %d batches back to back, each taken from HOST memory inside the timed region -- H2D of batch i+1 and the record arithmetic "
                                                     "of batch i-1 under the kernels of batch i, pipeline fill and drain included (one pipeline; the command line of value_e2e runs two per device)" % ks}
            out["value_streamed"] = out["value_upload_inclusive"]          # (round 3's name for it)
    th.close()
    if ticket:
        lib.th_ticket_close(ticket, 1 if rank == 0 else 0)
    if rank == 0:
        if e2e is not None:
            out["value_e2e"] = e2e
        if world == 1 and a.cpu_sample != 0:
            n_cpu = a.cpu_sample if a.cpu_sample > 0 else min(8192, max(256, 400 * effective_cpus()))      # ~15 s of oracle time at ~28 reads/s per core
            try:
                out["cpu_baseline"] = cpu_baseline(read_at, n_cpu, cfg)
            except Exception as ex:
                out["cpu_baseline"] = {"error": repr(ex)[:400]}
        others = a.others if a.others is not None else ("c2,c4,c5,c_long" if (a.config == "c3" and world == 1 and "TH_BENCH_LIB" not in os.environ) else "")
        if others:
            out["other_shapes"] = other_shapes([c for c in others.split(",") if c])
        # the figures a reader wants first, flat and LAST in the line (the driver keeps the tail of stdout)
        def g(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        kstep = g(out, "roofline", "kernel_ms_per_step") or {}
        nfail = [out.get("n_jobs_without_consensus"), g(out, "value_e2e", "n_jobs_without_consensus")] + [g(out, "other_shapes", c, "n_jobs_without_consensus") for c in ("c2", "c4", "c5", "c_long")]
        summ = {"value": out["value"], "value_upload_inclusive": g(out, "value_upload_inclusive", "value"), "value_e2e": g(out, "value_e2e", "value"),
                "e2e_cold": g(out, "value_e2e", "cold_reads_per_s"), "e2e_no_pause": g(out, "value_e2e", "run_without_pause", "reads_per_s"),
                "e2e_startup_s": g(out, "value_e2e", "runs", ) and [round(r["startup_s"], 2) for r in out["value_e2e"]["runs"]],
                "host_cpu_s_per_1M_reads": g(out, "value_e2e", "host_cpu_s_per_1M_reads"),
                "c2_reads_per_s": g(out, "other_shapes", "c2", "reads_per_s"), "c4_reads_per_s": g(out, "other_shapes", "c4", "reads_per_s"), "c5_reads_per_s": g(out, "other_shapes", "c5", "reads_per_s"),
                "c2_reads_per_step": g(out, "other_shapes", "c2", "reads_per_step"),
                "c_long_reads_per_s": g(out, "other_shapes", "c_long", "reads_per_s"), "c_long_mbases_per_s": g(out, "other_shapes", "c_long", "mbases_per_s"),
                "c_long_hits_ms": g(out, "other_shapes", "c_long", "kernel_ms_per_step", "th_k_encode+th_k_hits"), "c_long_ms_per_step": g(out, "other_shapes", "c_long", "ms_per_step"),
                "n_jobs_without_consensus": sum(x for x in nfail if x), "n_jobs_without_consensus_covers": "timed steps + warm-up, the end-to-end runs, the other shapes",
                "poa_ms": kstep.get("th_k_poa"), "chain_dp_ms": kstep.get("th_k_chain_dp"), "align_ms": kstep.get("th_k_align"), "hits_ms": kstep.get("th_k_encode+th_k_hits"),
                "chain_extract_ms": kstep.get("th_k_chain_extract"), "partition_ms": kstep.get("th_k_partition"), "ms_per_step": out["ms_per_step"],
                "c4_align_ms": g(out, "other_shapes", "c4", "kernel_ms_per_step", "th_k_align"), "c4_poa_ms": g(out, "other_shapes", "c4", "kernel_ms_per_step", "th_k_poa"),
                "c5_poa_ms": g(out, "other_shapes", "c5", "kernel_ms_per_step", "th_k_poa"), "c5_align_ms": g(out, "other_shapes", "c5", "kernel_ms_per_step", "th_k_align"),
                "roofline_frac": g(out, "roofline", "frac"), "roofline_frac_measured": g(out, "roofline", "frac_measured"), "traffic_stale": g(out, "roofline", "traffic_stale"),
                "issue_frac_poa": g(out, "roofline", "secondary", "kernels", "th_k_poa", "issue_frac"), "issue_frac_align": g(out, "roofline", "secondary", "kernels", "th_k_align", "issue_frac"),
                "issue_frac_chain_dp": g(out, "roofline", "secondary", "kernels", "th_k_chain_dp", "issue_frac"), "issue_instr_stale": g(out, "roofline", "secondary", "instr_stale"),
                "total_gcells_per_s": g(out, "roofline", "secondary", "total_gcells_per_s"), "poa_gcells_per_s": g(out, "roofline", "secondary", "kernels", "th_k_poa", "gcells_per_s"),
                "poa_valu_lane_instr_per_cell": g(out, "roofline", "secondary", "kernels", "th_k_poa", "valu_lane_instr_per_cell"),
                "cpu_baseline_reads_per_s": g(out, "cpu_baseline", "value"), "cpu_baseline_cores": g(out, "cpu_baseline", "cores")}
        summ = {k: (round(v, 4 if abs(v) < 10 else 1) if isinstance(v, float) else v) for k, v in summ.items()}
        out["summary"] = summ              # (dicts keep insertion order: this is the end of the line)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if a.config == "c5":
        for fn in kw["adapters"]:
            os.unlink(fn)
        os.rmdir(os.path.dirname(kw["adapters"][0]))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
