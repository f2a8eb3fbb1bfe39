#!/bin/bash
# Average shader clock and cycles per launch = GRBM_GUI_ACTIVE (one --pmc pass, kernel trace only).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/w4clock
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p -o p --output-format csv -- python3 $R/tools/w4_clock.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
cc = glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('$OUT/p/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
g = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    d = dur.get(r['Dispatch_Id'])
    if d and ('dense_bf16' in d[1] or 'Cijk' in d[1]):
        g.setdefault(d[1][:70], []).append((float(r['Counter_Value']), d[0]))
for k, v in g.items():
    v = v[5:]
    cyc = sum(c for c, _ in v) / len(v); ns = sum(t for _, t in v) / len(v)
    print('%-70s n=%2d  %.0f us  %.0f cycles  %.2f GHz  %.0f cycles/K-tile' % (k, len(v), ns / 1e3, cyc, cyc / ns, cyc / 128))
PY
