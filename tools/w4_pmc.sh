#!/bin/bash
# Developer tool: one --pmc pass (kernel trace only) over tools/w4_clock.py, averaged per kernel.
#   gpurun -- bash tools/w4_pmc.sh TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/w4pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/p -o p --output-format csv -- python3 $R/tools/w4_clock.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
cc = glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True)[0]
g = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    k = r['Kernel_Name'][:60]
    if 'dense_bf16' in k or 'Cijk' in k:
        g.setdefault(k, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, d in g.items():
    print(k)
    for c, v in d.items():
        v = v[5:] if len(v) > 10 else v
        print('    %-28s %14.0f per launch (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
