#!/bin/bash
# Developer tool: per-kernel average times of bench.py (rocprofv3 --kernel-trace --stats) under the
# default library and variant builds (tools/mkvar.sh names, or "default"), on one box.
#   gpurun -- bash tools/bench_kernels_ab.sh default oldw4
R=$PWD
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  if [ "$l" == default ]; then unset GNLP_LIB_PATH; else export GNLP_LIB_PATH=$R/tools/build/var/lib_$l.so; fi
  O=$R/gpurun_out/bk_ab/$l
  rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/log.txt || { echo "$l failed"; tail -3 $O/log.txt; exit 1; }
  echo "== $l"
  python3 - $O <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:8]:
    print('  %-60s calls %5s avg %9.1f us  %5.2f %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
d = json.loads(open(sys.argv[1] + '/bench.json').read().strip().splitlines()[-1])
print('  bench: %.1f seq/s' % d['value'])
PY
done
