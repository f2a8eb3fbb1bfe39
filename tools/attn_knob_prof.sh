#!/bin/bash
# Developer tool: KERNEL time (rocprofv3 --kernel-trace) of the fused attention kernel at the
# BERT-large shape under the values of one environment knob.  bash tools/attn_knob_prof.sh KNOB v1 v2 ...
R=$PWD
KNOB=$1; shift
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2; do
  for v in "$@"; do
    export $KNOB=$v
    O=$R/gpurun_out/attn_knob/$v.$rnd
    rm -rf $O; mkdir -p $O
    ATTN_ONE_ITERS=30 timeout -s KILL 90 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python $R/tools/attn_one.py > $O/log.txt 2>&1 || { echo "$v failed"; tail -3 $O/log.txt; continue; }
    python3 - "$KNOB=$v" $O <<'PY'
import csv, glob, sys
d = []
for f in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
    d += [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'attn_fwd' in r['Kernel_Name']]
d = sorted(d[3:]) if len(d) > 6 else sorted(d)
print('%-18s attn kernel: median %.1f us  min %.1f  (%d launches)' % (sys.argv[1], d[len(d) // 2] / 1e3, d[0] / 1e3, len(d)))
PY
  done
done
