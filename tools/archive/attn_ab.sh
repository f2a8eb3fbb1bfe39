#!/bin/bash
# Developer tool: the fused attention kernel's time at the BERT-large shape under variant builds
# of the library (tools/mkvar.sh), alternating, two rounds.   bash tools/archive/attn_ab.sh name1 name2 ...
for rnd in 1 2; do
  for l in "$@"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$PWD/tools/build/var/lib_$l.so; fi
    printf "%-12s " "$l"
    timeout -k 10 120 python - <<'PY' 2>&1 | tail -1
import sys; sys.path.insert(0, 'tools')
import attn_bench
ms, tf = attn_bench.run(64, 512, 16, iters=50)
ms2, _ = attn_bench.run(64, 512, 16, 64, False, True, iters=50)
print('bert_large B64 L512  %.1f us  %.0f TF   ragged %.1f us' % (ms * 1e3, tf, ms2 * 1e3))
PY
  done
done
