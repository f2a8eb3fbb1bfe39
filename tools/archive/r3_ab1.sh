#!/bin/bash
# Round 3, first A/B: single-issue softmax (+ row sums on the matrix pipe) and the polynomial GELU.
set -e
mkdir -p gpurun_out/r3_ab1
V=tools/build/var
for l in attn_v1 attn_v2; do
  echo "== tests with $l"
  GNLP_LIB_PATH=$PWD/$V/lib_$l.so timeout -k 10 400 python -u -m pytest tests/test_gpu_encoder.py tests/test_gpu_t5_decoder.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
done
for l in gelu7 gelu6; do
  echo "== tests with $l"
  GNLP_LIB_PATH=$PWD/$V/lib_$l.so timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "dense or gelu or act" --timeout 120 --timeout-method thread 2>&1 | tail -3
done
for l in default attn_v1 attn_v2; do
  echo "== attn_bench $l"
  if [ $l == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/$V/lib_$l.so; fi
  timeout -k 10 200 python tools/attn_bench.py 2>&1 | tail -8
done
unset GNLP_LIB_PATH
echo "== bench A/B"
bash tools/ab_lib.sh $V/lib_attn_v1.so $V/lib_attn_v2.so $V/lib_gelu0.so $V/lib_gelu7.so $V/lib_gelu6.so
