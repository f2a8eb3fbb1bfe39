#!/bin/bash
# Developer tool: KERNEL time (rocprofv3 --kernel-trace) of the fused attention kernel at the
# BERT-large shape under variant builds of the library (tools/mkvar.sh); wall-clock loops from
# Python are host-bound at ~100 us per call.   bash tools/attn_ab_prof.sh name1 name2 ...
R=$PWD
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2; do
  for l in "$@"; do
    if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$R/tools/build/var/lib_$l.so; fi
    O=$R/gpurun_out/attn_ab/$l.$rnd
    rm -rf $O; mkdir -p $O
    ATTN_ONE_ITERS=30 timeout -s KILL 90 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python $R/tools/attn_one.py > $O/log.txt 2>&1 || { echo "$l failed"; tail -3 $O/log.txt; continue; }
    python3 - "$l" $O <<'PY'
import csv, glob, sys
d = []
for f in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
    d += [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'attn_fwd' in r['Kernel_Name']]
d = sorted(d[3:]) if len(d) > 6 else sorted(d)
print('%-12s attn kernel: median %.1f us  min %.1f  (%d launches)' % (sys.argv[1], d[len(d) // 2] / 1e3, d[0] / 1e3, len(d)))
PY
  done
done
