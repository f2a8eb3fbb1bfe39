#!/bin/bash
# Developer tool: one GPU call for the one-wave-per-SIMD GEMM experiments.  Every GPU step has its
# own time limit and the script stops at the first step that fails or times out.
set -e
set -o pipefail
mkdir -p gpurun_out
for w in 1 2; do for b in 0 1; do
  echo "wps=$w big=$b" >> gpurun_out/mfma_peak.log
  timeout -k 5 30 tools/build/mfma_peak $w 1 4000 10 $b | tail -2 >> gpurun_out/mfma_peak.log
done; done
cat gpurun_out/mfma_peak.log
for v in 30 40 50 60; do
  GNLP_W4_ABL=$v GNLP_GEMM_TILE=256 BENCH=0 timeout -k 10 100 python -u tools/w4_check.py > gpurun_out/w4_check$v.log 2>&1 || { tail -5 gpurun_out/w4_check$v.log; echo "variant $v FAILED"; exit 1; }
  echo "variant $v: $(grep -E 'ALL OK|MISMATCH' gpurun_out/w4_check$v.log)"
  grep -q "ALL OK" gpurun_out/w4_check$v.log || { grep identical gpurun_out/w4_check$v.log; }
done
timeout -k 10 240 python -u tools/w4_abl.py > gpurun_out/w4_abl.log 2>&1
cat gpurun_out/w4_abl.log
