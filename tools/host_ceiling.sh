#!/bin/bash
# GPU box: what the HOST side of the command line sustains when the device stage costs nothing -- the eight-GPU host budget measured on the one box there is.
# TH_REPLAY=1 (th_pipeline.hip, measurement only): a pipeline's first batch runs for real, every later batch of the same size gets that batch's result records copied back instead
# of kernels.  The input is K copies of ONE set of S reads with batch = S and no ramp step, so every batch holds the same reads and the replayed output is still the right bytes:
# its md5 is compared with a real run of the same file.  Reader -> upload (real H2D) -> record arithmetic + formatting -> writer then run with P pipelines (all on device 0) at
# whatever rate the host sustains under the container's CPU quota; TH_TRACE prints the steady rate (per pipeline, after its first batch) and the CPU seconds by role.
# REPLAY_ENV="TH_WAVES_POA=768 TH_WAVES_ALN=1024" (replay runs only): smaller persistent grids, so that EIGHT pipelines' POA scratch fits the one device's memory.
#   usage: tools/host_ceiling.sh [outfile=gpurun_out/host_ceiling.txt] [pipelines="8 4 2 1"] [S=16384] [K=122]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${1:-$R/gpurun_out/host_ceiling.txt}; PIPES=${2:-8 4 2 1}; S=${3:-16384}; K=${4:-122}
D=/dev/shm; [ -d $D ] || D=/tmp
cd $R; mkdir -p $(dirname $O); : > $O
python3 - <<PY >> $O
import sys, time; sys.path.insert(0, "$R")
import tidehunter_amd as TH
lib = TH.load_library()
reads = [TH.synth_read(lib, 1, i, 1000, 10, 0.10) for i in range($S)]
with open("$D/hc.fa", "wb") as f:
    for c in range($K):
        f.write(b"".join(b">syn_%d\n" % (c * $S + i) + s + b"\n" for i, s in enumerate(reads)))
print("input: %d copies of a set of %d headline reads = %d reads, %.1f GB of FASTA in $D" % ($K, $S, $K * $S, sum(len(s) + 12 for s in reads) * $K / 1e9))
PY
nproc >> $O; cat /sys/fs/cgroup/cpu.max >> $O 2>/dev/null; free -g | head -2 >> $O
g() { local n=$1 l=0; for ((i = 1; i < n; ++i)); do l="$l,0"; done; echo $l; }
echo "== real run, 1 pipeline (the bytes the replays must reproduce)" >> $O
sleep 6
TH_WORKERS_PER_DEV=1 TH_RAMP0=$S TH_TRACE=1 $R/tidehunter_amd/TideHunter-amd -b $S -o $D/hc.real $D/hc.fa 2>&1 | grep "run_file\]\|Real time" >> $O
REAL=$(md5sum < $D/hc.real | cut -d' ' -f1); echo "records: $(grep -c '^>' $D/hc.real) md5: $REAL" >> $O; rm -f $D/hc.real
for P in $PIPES; do
  sleep 6
  echo "== TH_REPLAY=1, $P pipeline(s) on device 0, batch $S" >> $O
  env ${REPLAY_ENV:-TH_X=0} TH_REPLAY=1 TH_WORKERS_PER_DEV=1 TH_RAMP0=$S TH_TRACE=1 timeout 600 $R/tidehunter_amd/TideHunter-amd -g $(g $P) -b $S -o $D/hc.out $D/hc.fa 2>&1 | grep "run_file\]\|Real time\|rror" >> $O
  M=$(md5sum < $D/hc.out | cut -d' ' -f1); echo "records: $(grep -c '^>' $D/hc.out) md5: $M $([ "$M" = "$REAL" ] && echo '(= the real run)' || echo '(DIFFERS from the real run)')" >> $O
  rm -f $D/hc.out
done
rm -f $D/hc.fa
cat $O
