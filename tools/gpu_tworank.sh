#!/bin/bash
# development helper (GPU box): bench.py's own multi-rank path on hardware -- two ranks (gloo) sharing the one GPU of the box.
# Not a scaling measurement (both ranks share a device); it shows that the rank path, the partition by bases, the ticket
# counter and the result digest work with the real library.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-tworank}; mkdir -p $O; cd $R
export TH_BENCH_BACKEND=gloo TH_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1
MASTER_PORT=29601 timeout 900 python bench.py --gpus 1 --batch 40000 --steps 1 --warmup 1 --digest --cpu-sample 0 --e2e 0 > $O/one.json 2> $O/one.err
MASTER_PORT=29602 timeout 900 python bench.py --gpus 2 --batch 20000 --steps 1 --warmup 1 --digest --cpu-sample 0 --e2e 0 > $O/two.json 2> $O/two.err
MASTER_PORT=29603 timeout 900 python bench.py --gpus 2 --batch 20000 --steps 1 --warmup 1 --digest --cpu-sample 0 --e2e 0 --steal 4 > $O/steal.json 2> $O/steal.err
MASTER_PORT=29604 timeout 900 python bench.py --gpus 2 --config c5 --batch 10000 --steps 1 --warmup 1 --digest --cpu-sample 0 --e2e 0 > $O/two_c5.json 2> $O/two_c5.err
MASTER_PORT=29605 timeout 900 python bench.py --gpus 1 --config c5 --batch 20000 --steps 1 --warmup 1 --digest --cpu-sample 0 --e2e 0 > $O/one_c5.json 2> $O/one_c5.err
python3 - <<PY
import json
for n in ("one", "two", "steal", "one_c5", "two_c5"):
    try:
        j = json.load(open("$O/%s.json" % n)); print(n, "n_gpus", j["n_gpus"], "reads/s", round(j["value"]), "digest", j["records_digest"], j["records_digest_reads"], j["config"]["sharding"][:60])
    except Exception as e:
        print(n, "FAILED", e); print(open("$O/%s.err" % n).read()[-800:])
PY
