#!/bin/bash
# Average shader clock per kernel = GRBM_GUI_ACTIVE / duration (one --pmc pass, kernel trace only).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/clock
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p -o p --output-format csv -- python $R/tools/archive/gemm_power.py > $OUT/log.txt 2>&1 || tail -5 $OUT/log.txt
python - <<PY
import csv, glob, collections
cc = glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('$OUT/p/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
rows = []
for r in csv.DictReader(open(cc)):
    d = dur.get(r['Dispatch_Id'])
    if d and 'dense_bf16_' in d[1]:
        rows.append((int(r['Dispatch_Id']), float(r['Counter_Value']), d[0]))
rows.sort()
# gemm_power.py: per kind and shape 35 launches
for g in range(0, len(rows), 35):
    grp = rows[g:g + 35][5:]
    if not grp: continue
    cyc = sum(c for _, c, _ in grp) / len(grp); ns = sum(t for _, _, t in grp) / len(grp)
    print('launch group %2d: %.0f us  GRBM_GUI_ACTIVE %.0f  -> %.2f GHz' % (g // 35, ns / 1e3, cyc, cyc / ns))
PY
