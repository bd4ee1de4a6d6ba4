#!/usr/bin/env python3
"""bench.py -- reads/s of the TideHunter hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (encode -> self-hits -> chain -> partition -> POA ->
unit/consensus alignments -> host record arithmetic) over one batch of synthetic reads of the
headline configuration (BASELINE.json configs[2]: period 1000 x 10 copies, 10 % error, ~10 kb reads).
The batch is uploaded to HBM before the timed region (inputs resident), every step processes a
different batch (read index = (rank_offset + step * batch + i) mod 100000); the default batch is the whole 100k-read set.

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; reads shard across ranks
(no data-path collective; RCCL is used for the barrier and the max-over-ranks of the time only).

Prints ONE JSON line (rank 0) with `roofline` (the kernel with the largest device time -- th_k_poa --, HIP-event
timed on the kernel's own stream inside the library) and `cpu_baseline` (the CPU oracle, a scalar C port of the
reference's algorithm, timed on the host cores on a bounded sample of the same workload; N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PERIOD, COPIES, ERR, SEED = 1000, 10, 0.10, 1
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(lib, TH, n_reads):
    """CPU oracle (oracle/tho_oracle, kind 'port') on the first n_reads of the same read set."""
    import multiprocessing
    cores = multiprocessing.cpu_count()
    exe = os.path.join(ROOT, "oracle", "tho_oracle")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "tho_oracle"], check=True)
    path = "/tmp/th_bench_cpu_sample_%d.fa" % os.getpid()
    with open(path, "wb") as f:
        for i in range(n_reads):
            f.write(b">syn_p%d_c%d_e%.2f_s%d_%d\n" % (PERIOD, COPIES, ERR, SEED, i) + TH.synth_read(lib, SEED, i, PERIOD, COPIES, ERR) + b"\n")
    t0 = time.time()
    subprocess.run([exe, "-t", str(cores), "-o", "/dev/null", path], check=True, stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    os.unlink(path)
    return {"value": n_reads / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%d reads of the same set (period %d x %d, %.0f%% err), oracle/tho_oracle -t %d, %.1f s"
                      % (n_reads, PERIOD, COPIES, ERR * 100, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)   # two: device buffers reach their steady size (a batch that grows one is first-touched inside a kernel)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TH_BENCH_BATCH", "100000")), help="reads per step per GPU")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1: auto, 0: skip)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # TH_BENCH_BACKEND / TH_BENCH_DEVICE exist to rehearse the multi-rank path on a one-GPU box (gloo, every rank on device 0)
    backend = os.environ.get("TH_BENCH_BACKEND", "nccl")
    if "TH_BENCH_DEVICE" in os.environ:
        local = int(os.environ["TH_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    import tidehunter_amd as TH
    lib = TH.load_library()
    th = TH.TideHunter(device=local)

    B = a.batch
    n_batches = a.warmup + a.steps

    def make_batch(b):
        base = (rank * n_batches + b) * B
        return [TH.synth_read(lib, SEED, (base + i) % 100000, PERIOD, COPIES, ERR) for i in range(B)]

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    batches = [make_batch(b) for b in range(n_batches)]
    bases = sum(len(s) for bt in batches[a.warmup:] for s in bt)
    for b in range(a.warmup):
        th.upload(batches[b]); th.run(); th.collect()
    # timed region: inputs are made resident first (upload outside), then run + collect are timed
    t_total, n_cons, alg_bytes, kms = 0.0, 0, 0, {}
    for b in range(a.warmup, n_batches):
        th.upload(batches[b])
        sync_all()
        t0 = time.perf_counter()
        th.run()                 # all kernels + the small D2H of result records
        th.collect_native()      # host-side record arithmetic in C (src/gen_cons.c:208-301)
        torch.cuda.synchronize()
        t_total += time.perf_counter() - t0
        recs = th.records()      # Python dict view for the byte accounting below: not part of the path
        for kname, v in th.timing().items():
            kms[kname] = kms.get(kname, 0.0) + v
        for s, rr in zip(batches[b], recs):
            alg_bytes += len(s)                                     # B_alg = L + sum(cons_len + 4*pos_n + 40)
            for r in rr:
                alg_bytes += r["cons_len"] + 4 * len(r["sub_pos"]) + 40
                n_cons += 1
    sync_all()
    if dist is not None:
        t = torch.tensor([t_total], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_total = float(t.item())
    reads = B * a.steps * world
    value = reads / t_total
    out = {
        "metric": "reads/sec on 10 kb period-1000x10 reads (TideHunter hot path)", "value": value, "unit": "reads/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * t_total / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "mbases_per_s": bases * world / t_total / 1e6,
        "config": {"workload": "C3 headline: synthetic tandem-repeat reads, period 1000 x 10 copies, 10 percent error (sub/ins/del), ~10 kb; "
                               + "%d reads per step per GPU drawn from the 100k-read set (seed 1)" % B,
                   "reads_per_step_per_gpu": B, "sharding": "contiguous read blocks per rank, no data-path collective"},
        "consensus_per_read": n_cons / float(B * a.steps),
    }
    if rank == 0:
        dom = max(kms, key=kms.get)                      # dominant kernel = largest device time
        launch_s = kms[dom] / 1000.0 / a.steps             # one launch of it per step
        per_launch_bytes = alg_bytes / float(a.steps)
        ach = per_launch_bytes / launch_s / 1e9 if launch_s > 0 else 0.0
        # traffic: HBM bytes of one launch of the dominant kernel from the PMC passes of this same command
        # (tools/prof_round.sh -> profiles/r01_hbm_traffic.json); counters cannot be read from inside the process, so
        # the committed profile is quoted when it was taken at the same batch size, else null
        traffic, traffic_src = None, None
        try:
            prof = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic.json")))
            if prof.get("reads_per_launch") == B:
                for kname, v in prof["kernels"].items():
                    if kname.replace("void ", "").split("<")[0] == dom:
                        traffic, traffic_src = v["hbm_bytes_per_launch_corrected"], "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_src, "launch_ms": launch_s * 1000.0,
                           "kernel_ms_per_step": {k: v / a.steps for k, v in kms.items()},
                           "algorithmic_bytes_per_launch": per_launch_bytes}
        if world == 1 and a.cpu_sample != 0:
            n_cpu = a.cpu_sample if a.cpu_sample > 0 else min(4096, max(64, 16 * (os.cpu_count() or 8)))
            out["cpu_baseline"] = cpu_baseline(lib, TH, n_cpu)
        print(json.dumps(out))
    th.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
