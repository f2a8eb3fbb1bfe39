#!/bin/bash
# Developer tool: bench.py with the LayerNorm fold off / on, alternating, under the default library
# or variant builds (tools/mkvar.sh).   bash tools/archive/fold_ab.sh [name ...]
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=(default)
for rnd in 1 2; do
  for l in "${LIBS[@]}"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/tools/build/var/lib_$l.so; fi
    for k in 0 1; do
      GNLP_LN_FOLD=$k timeout -k 10 150 python bench.py --no-cpu-baseline > gpurun_out/fab.json 2> gpurun_out/fab.err
      python3 -c "
import json; d = json.load(open('gpurun_out/fab.json'))
print('%-12s fold=$k  %.1f seq/s  %.3f ms/step  FFN-1 %.4f ms' % ('$l', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
    done
  done
done
