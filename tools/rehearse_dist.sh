#!/bin/bash
# Rehearsal of bench.py's multi-rank control flow on a 1-GPU box: launched the way the driver
# launches N > 1 (torch.distributed.run), one rank, RCCL group forced on.
set -e
mkdir -p gpurun_out
GNLP_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 \
  > gpurun_out/rehearse_dist.json 2> gpurun_out/rehearse_dist.err || { tail -30 gpurun_out/rehearse_dist.err; exit 1; }
tail -1 gpurun_out/rehearse_dist.json
