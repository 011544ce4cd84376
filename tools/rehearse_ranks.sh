#!/bin/bash
# Rehearsal of `bench.py --gpus N` on ONE GPU: N processes share the card, the collectives run over gloo (the kernels,
# the plans, the candidate loop and the JSON line are the real ones; the times say nothing about xGMI).
# usage: tools/rehearse_ranks.sh <N> <out.json> [bench args...]
N=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
VFM_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port $((29500 + N)) $R/bench.py --gpus $N --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT 2> ${OUT%.json}.err
