#!/bin/bash
# Developer tool: kernel times of tools/bwd_one.py under variant builds of the library
#   gpurun -- bash tools/archive/bwd_exp.sh tools/build/a/libgluonnlp_amd.so tools/build/b/libgluonnlp_amd.so
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/bwd_exp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
i=0
for l in default "$@"; do
  i=$((i+1))
  if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$R/$l; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $OUT/p$i -o x --output-format csv -- python $R/tools/bwd_one.py > $OUT/p$i.log 2>&1 || { echo "run $i failed" >> $OUT/summary.txt; tail -5 $OUT/p$i.log >> $OUT/summary.txt; exit 1; }
  f=$(find $OUT/p$i -name 'x_kernel_stats.csv' | head -1)
  grep attn_bwd $f | cut -d, -f1-4 | sed "s|^|$l |" >> $OUT/summary.txt
done
cat $OUT/summary.txt
