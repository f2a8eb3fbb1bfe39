#!/bin/bash
# Developer tool (round 2): forward-only A/B of several library builds (3 rounds, alternating),
# with a quick dense-correctness check of each build first.
set -e
mkdir -p gpurun_out
for l in "$@"; do
  BENCH=0 GNLP_LIB_PATH=$PWD/$l timeout -k 10 200 python tools/w4_check.py 2>&1 | grep -E "ALL OK|MISMATCH" | sed "s|^|$l: |" || true
done
bash tools/ab_lib.sh "$@"
