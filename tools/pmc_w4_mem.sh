#!/bin/bash
# Developer tool: memory-side counters of one persistent-GEMM shape (tools/gemm_one.py) under the
# default library and variant builds; one --pmc pass per run.
#   SHAPE=32768,3072,1024 gpurun -- bash tools/pmc_w4_mem.sh default oldw4
R=$PWD
cd /tmp && export TMPDIR=/tmp
export SHAPE=${SHAPE:-32768,3072,1024} ACT=${ACT:-identity} ITERS=5
for l in "$@"; do
  if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$R/tools/build/var/lib_$l.so; fi
  i=0
  for C in "WRITE_SIZE" "FETCH_SIZE" "TCC_WRITE_sum TCC_READ_sum TCC_MISS_sum TCC_EA_WRREQ_sum"; do
    i=$((i+1))
    O=$R/gpurun_out/pmc_w4_mem/$l.$i
    rm -rf $O; mkdir -p $O
    timeout -s KILL 90 rocprofv3 --pmc $C --kernel-trace -d $O -o p --output-format csv -- python $R/tools/gemm_one.py > $O/log.txt 2>&1 || { echo "$l pass $i failed"; tail -3 $O/log.txt; continue; }
    python3 - $l $O <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[2] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dense_bf16_w4' not in r['Kernel_Name']: continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print(sys.argv[1], ' '.join('%s=%.4g' % (k, v / max(n, 1)) for k, (v, n) in sorted(agg.items())))
PY
  done
done
