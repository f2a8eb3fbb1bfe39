#!/bin/bash
# Developer tool (round 2): correctness of the dense tests on the default build, then the isolated
# GEMM shapes and the forward for the default build against variant libraries, alternating.
#   gpurun -- bash tools/archive/r2_ab.sh tools/build/base/libgluonnlp_amd.so
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py -q -x -k "dense" --timeout 200 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
for rnd in 1 2; do
  for l in default "$@"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/$l; fi
    echo "== $l"; ROUNDS=1 timeout -k 10 200 python tools/gemm_bench.py
  done
done
unset GNLP_LIB_PATH
bash tools/ab_lib.sh "$@"
