#!/bin/bash
# Developer tool: the 256x192-tile dense kernel on the U = 768 configurations, tests then A/B.
#   gpurun -- bash tools/archive/ab_nt.sh
set -e
mkdir -p gpurun_out
bash tools/ab_bench.sh GNLP_GEMM_NT 256 auto 192 -- --config t5_base --ragged
bash tools/ab_bench.sh GNLP_GEMM_NT 256 auto -- --config bert_base --ragged
