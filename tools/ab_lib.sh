#!/bin/bash
# Developer tool: A/B of variant builds of the library on bench.py (GNLP_LIB_PATH), alternating.
#   gpurun -- bash tools/ab_lib.sh tools/build/var/lib_x.so tools/build/var/lib_y.so [-- bench args]
set -e
LIBS=(default)
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for rnd in 1 2 3; do
  for l in "${LIBS[@]}"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/$l; fi
    timeout -k 10 150 python bench.py --no-cpu-baseline "$@" > gpurun_out/abl.json 2> gpurun_out/abl.err
    python3 -c "
import json; d = json.load(open('gpurun_out/abl.json'))
print('%-32s %.1f seq/s  %.3f ms/step  dominant kernel %.4f ms' % ('$l', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
