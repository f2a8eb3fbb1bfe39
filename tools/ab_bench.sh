#!/bin/bash
# Developer tool: in-situ A/B of an environment knob on bench.py (the forward, not back-to-back
# launches: the chip's clock depends on the mix).  usage: ab_bench.sh KNOB v1 v2 ... [-- bench args]
set -e
KNOB=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for rnd in 1 2; do
  for v in "${VALS[@]}"; do
    env $KNOB=$v timeout -k 10 150 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    python3 -c "
import json; d = json.load(open('gpurun_out/ab_$v.json'))
print('$KNOB=$v  %.1f seq/s  %.3f ms/step  dominant kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
