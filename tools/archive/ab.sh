#!/bin/bash
# A/B of library builds on one box: tools/archive/ab.sh libA.so libB.so ...  (interleaved, 2 rounds)
for r in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    GNLP_LIB_PATH=$PWD/$lib ROUNDS=1 timeout -k 10 150 python tools/gemm_bench.py 2>&1 | tail -1 || exit 1
  done
done
