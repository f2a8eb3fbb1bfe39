#!/bin/bash
# Developer tool (round 2): dense tests, then bench.py per configuration for the default build
# against variant libraries (2 rounds, alternating).  usage: r2_cfg.sh lib.so ... 
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py -q -x -k "dense" --timeout 200 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
for cfg in "bert_large" "bert_base --ragged" "t5_base --ragged"; do
  for rnd in 1 2; do
    for l in default "$@"; do
      if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/$l; fi
      timeout -k 10 150 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/abl.json 2> gpurun_out/abl.err
      python3 -c "
import json; d = json.load(open('gpurun_out/abl.json'))
print('%-14s %-38s %.1f seq/s  %.3f ms/step  frac %.4f' % ('$cfg'[:14], '$l', d['value'], d['ms_per_step'], d['path_frac_of_mfma_peak']))"
    done
  done
done
