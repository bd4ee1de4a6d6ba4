"""bench.py's OWN multi-rank path, rehearsed without GPUs: launched under torch.distributed.run with two ranks, gloo, and the
CPU-emulation build of the kernels (TH_BENCH_LIB).  Asserts the contract of the JSON line (one line, n_gpus = 2) and that the
sharded run -- static partition by cumulative bases, and blocks stolen through the shared ticket counter -- produces exactly
the records of the single-rank run over the same global read set."""
import json
import os
import subprocess
import sys

import th_util as U

ROOT = U.ROOT


def _bench(tmp_path, nproc, port, *args, self_spawn=False):
    env = dict(os.environ, TH_BENCH_BACKEND="gloo", TH_BENCH_LIB=U.emul_lib_path(), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--config", "tiny", "--steps", "1", "--warmup", "1", "--digest", "--cpu-sample", "0", "--e2e", "0", *args]
    if nproc == 1 or self_spawn:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *common]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *common]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_two_ranks_match_one_rank(tmp_path):
    one = _bench(tmp_path, 1, 29541, "--batch", "8")
    two = _bench(tmp_path, 2, 29542, "--batch", "4")
    steal = _bench(tmp_path, 2, 29543, "--batch", "4", "--steal", "2")
    spawned = _bench(tmp_path, 2, 29544, "--batch", "4", self_spawn=True)      # `python bench.py --gpus 2` without a launcher starts one itself
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and steal["n_gpus"] == 2 and spawned["n_gpus"] == 2
    assert spawned["records_digest"] == one["records_digest"]
    for j in (one, two, steal):
        assert j["unit"] == "reads/s" and j["value"] > 0 and j["scaling"] == "weak" and j["records_digest_reads"] == 8
        assert "roofline" in j and j["roofline"]["bound"] == "hbm"
    assert one["records_digest"] == two["records_digest"] == steal["records_digest"]
    assert two["config"]["inputs_resident_before_timing"] and not steal["config"]["inputs_resident_before_timing"]
    # the ordered result gather: the text of the two ranks' blocks, gathered to rank 0 in rank order, is the text of the one-rank run
    assert two["result_gather"]["backend"] == "gloo" and len(two["result_gather"]["bytes_by_rank"]) == 2 and all(b > 0 for b in two["result_gather"]["bytes_by_rank"])
    assert two["result_gather"]["output_md5"] == one["result_gather"]["output_md5"] == spawned["result_gather"]["output_md5"]
    assert two["result_gather"]["bytes"] == one["result_gather"]["bytes"] > 100


def test_bench_eight_ranks_rehearsal(tmp_path):
    """the launch the driver makes for SCALE (8 ranks under torch.distributed.run), rehearsed on the emulation library: static blocks
    and stolen blocks give the records of one rank over the same global list, every rank gets a block, one JSON line"""
    one = _bench(tmp_path, 1, 29551, "--batch", "16")
    eight = _bench(tmp_path, 8, 29552, "--batch", "2")
    steal = _bench(tmp_path, 8, 29553, "--batch", "2", "--steal", "2")
    assert eight["n_gpus"] == 8 and steal["n_gpus"] == 8
    assert one["records_digest"] == eight["records_digest"] == steal["records_digest"]
    assert eight["records_digest_reads"] == 16 and steal["records_digest_reads"] == 16
    assert eight["result_gather"]["output_md5"] == one["result_gather"]["output_md5"] and len(eight["result_gather"]["bytes_by_rank"]) == 8
    sys.path.insert(0, ROOT)
    import bench
    lens = [100 + 7 * (i % 5) for i in range(16)]
    assert all(b > a for a, b in zip(bench.partition_by_bases(lens, 8), bench.partition_by_bases(lens, 8)[1:]))      # no empty block


def test_partition_by_bases_balances_bases_not_counts():
    sys.path.insert(0, ROOT)
    import bench
    lens = [10000] * 40 + [15000] * 40          # config 5: the 5000-period class is 1.5 x longer
    cuts = bench.partition_by_bases(lens, 8)
    assert cuts[0] == 0 and cuts[-1] == 80 and all(a < b for a, b in zip(cuts, cuts[1:]))
    per = [sum(lens[a:b]) for a, b in zip(cuts, cuts[1:])]
    assert max(per) - min(per) <= 2 * 15000
    assert (cuts[4] - cuts[0]) > (cuts[8] - cuts[4])              # more (shorter) reads in the first half
    assert bench.partition_by_bases([5, 5, 5], 8)[-1] == 3      # fewer reads than parts: empty blocks allowed, nothing lost
    assert bench.partition_by_bases([], 4) == [0, 0]
