#!/bin/bash
# Developer tool: LayerNorm kernel variants inside the forward (rocprofv3 per-kernel average).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/abn
cd /tmp && export TMPDIR=/tmp
for mv in f a; do for nt in 0 1; do
  export GNLP_NORM_MAXV=$mv GNLP_NORM_NT=$nt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abn/$mv$nt -o s --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/abn/$mv$nt.json 2> $R/gpurun_out/abn/$mv$nt.err
  f=$(find $R/gpurun_out/abn/$mv$nt -name '*kernel_stats.csv' | head -1)
  echo "MAXV=$mv NT=$nt: $(grep norm_vec $f | cut -d, -f1-4 | sed 's/.*EEEv[^,]*,/ /')  $(python -c "import json;print('%.1f seq/s' % json.load(open('$R/gpurun_out/abn/$mv$nt.json'))['value'])")"
  rm -rf $R/gpurun_out/abn/$mv$nt
done; done
