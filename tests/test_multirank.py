"""N > 1 path: reads shard across ranks in contiguous blocks with no data-path collective; rank 0 can
gather the per-rank results in order.  World size 2, gloo, CPU (the emulation library stands in for
the HIP library; the sharding/gather logic is what is under test)."""
import os
import subprocess
import sys

import th_util as U

ROOT = U.ROOT

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import tidehunter_amd as TH
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
th = TH.TideHunter(lib_path=%(emu)r)
reads = [TH.synth_read(th.lib, 3, i, 60 + 10 * (i %% 3), 6, 0.08) for i in range(6)]
lo, hi = rank * len(reads) // world, (rank + 1) * len(reads) // world        # contiguous block per rank
mine = th.process(reads[lo:hi])
dist.barrier()
gathered = [None] * world
dist.all_gather_object(gathered, (lo, mine))
if rank == 0:
    merged = [r for _, part in sorted(gathered) for r in part]
    single = th.process(reads)
    assert merged == single, "sharded result differs from the single-rank result"
    print("MULTIRANK_OK", len(merged))
th.close()
dist.destroy_process_group()
'''


def test_two_rank_sharding_matches_single_rank(tmp_path):
    emu = U.emul_lib_path()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, emu=emu))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert "MULTIRANK_OK 6" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
