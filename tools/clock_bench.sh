#!/bin/bash
# Effective shader clock per kernel of bench.py (GRBM_GUI_ACTIVE / 8 XCDs / duration), one --pmc pass.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/clockb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1 || tail -5 $OUT/log.txt
python - <<PY
import csv, glob, collections
cc = glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('$OUT/p/**/*kernel_trace.csv', recursive=True)[0]
dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    d = dur.get(r['Dispatch_Id'])
    if not d or d[0] < 10000: continue
    n = d[1].replace('gnlp::', '')[:48]
    a = agg[n]; a[0] += float(r['Counter_Value']) / 8.0; a[1] += d[0]; a[2] += 1
for n, (c, t, k) in sorted(agg.items(), key=lambda x: -x[1][1])[:8]:
    print('%-50s launches %4d  avg %.0f us  %.0f kcycles  %.2f GHz' % (n, k, t / k / 1e3, c / k / 1e3, c / t))
PY
