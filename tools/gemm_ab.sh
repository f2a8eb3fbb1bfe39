#!/bin/bash
# Developer tool: tools/gemm_bench.py (the four GEMM shapes of a BERT-large layer, back to back)
# under variant builds of the library (tools/mkvar.sh), alternating.  bash tools/gemm_ab.sh name1 ...
for rnd in 1 2; do
  for l in "$@"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/tools/build/var/lib_$l.so; fi
    printf "%-12s " "$l"
    ROUNDS=1 timeout -k 10 120 python tools/gemm_bench.py 2>&1 | tail -1
  done
done
