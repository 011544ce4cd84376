#!/bin/bash
# Rehearsal of `bench.py --gpus N` on ONE GPU: N processes share the card, the collectives run over gloo (the kernels,
# the plans, the candidate loop, the time budget and the JSON line are the real ones; the times say nothing about xGMI).
# Run with the driver's own arguments (--steps 20 --warmup 5, everything else default); the wall time of the whole command
# is appended to <out>.wall -- the first-run safety figure (bench.py --total-budget-s, default 300 s).
# usage: tools/rehearse_ranks.sh <N> <out.json> [bench args...]
N=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
T0=$(date +%s.%N)
VFM_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port $((29500 + N)) $R/bench.py --gpus $N --steps 20 --warmup 5 "$@" > $OUT 2> ${OUT%.json}.err
RC=$?
T1=$(date +%s.%N)
echo "ranks=$N rc=$RC wall_s=$(python3 -c "print(round($T1 - $T0, 1))") args: --gpus $N --steps 20 --warmup 5 $*" | tee ${OUT%.json}.wall
