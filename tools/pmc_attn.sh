#!/bin/bash
# PMC passes over the fused attention kernel at the BERT-large shape (tools/attn_one.py); each
# pass is its own short run (counters only, no tracing domains beside --kernel-trace).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python $R/tools/attn_one.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'attn_fwd' not in r['Kernel_Name']: continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in agg.items():
        print(k, v / max(n, 1), n)
for f in sorted(glob.glob('$OUT/p1/**/*kernel_trace.csv', recursive=True)):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'attn_fwd' in r['Kernel_Name']]
    print('attn kernel ns', sum(d) / max(len(d), 1), len(d))
PY
