#!/bin/bash
# Where the K-split ("skinny") bf16 Dense kernel stops paying: BERT-base forward at small batches
# with the kernel allowed up to M rows (GNLP_SKINNY_MAX_M) against the tiled kernel.
set -e
cd $GRAFT_REPO_ROOT
for b in ${BATCHES:-1 4 8 16 32 64}; do
  for mx in 64 1048576; do
    GNLP_SKINNY_MAX_M=$mx timeout -k 10 120 python bench.py --config bert_base --batch $b --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/sk_${b}_${mx}.json 2> gpurun_out/sk.err
    python - <<PY
import json
j = json.loads(open('gpurun_out/sk_${b}_${mx}.json').read().strip().splitlines()[-1])
print('B=%-3d M=%-5d skinny_max_m=%-8d %.3f ms' % ($b, $b * 128, $mx, j['ms_per_step']), flush=True)
PY
  done
done
