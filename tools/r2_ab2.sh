#!/bin/bash
# Developer tool (round 2): isolated proj / ffn2 shapes + the forward for several library builds.
set -e
mkdir -p gpurun_out
for rnd in 1 2; do
  for l in default "$@"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/$l; fi
    echo "== $l"; ROUNDS=1 timeout -k 10 200 python tools/gemm_bench.py
  done
done
unset GNLP_LIB_PATH
bash tools/ab_lib.sh "$@"
