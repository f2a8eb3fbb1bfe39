#!/bin/bash
# Developer tool: SQ counter passes over one short script, averaged over the kernels whose name
# contains a substring; each pass is its own run (counters only, beside --kernel-trace).
#   gpurun -- bash tools/pmc_one.sh tools/normbwd_one.py norm_bwd
set -e
SCRIPT=$1
PAT=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_one
rm -rf $OUT && mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVES" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python $R/$SCRIPT > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
PAT=$PAT OUT=$OUT python - <<'PY'
import csv, glob, collections, os
pat, out = os.environ['PAT'], os.environ['OUT']
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(k, v / max(n, 1), n)
for f in sorted(glob.glob(out + '/p1/**/*kernel_trace.csv', recursive=True)):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
    print('kernel ns', sum(d) / max(len(d), 1), len(d))
PY
