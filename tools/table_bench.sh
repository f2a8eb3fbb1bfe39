#!/bin/bash
# Developer tool: the DESIGN.md result table (one bench.py run per BASELINE.md configuration).
set -e
mkdir -p gpurun_out/table
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/table/$name.json 2> gpurun_out/table/$name.err; python3 -c "
import json; d = json.load(open('gpurun_out/table/$name.json'))
print('%-28s %9.1f seq/s  %8.3f ms  path %.0f TF  %.1f %% of peak' % ('$name', d['value'], d['ms_per_step'], d['path_tflops'], 100 * d['path_frac_of_mfma_peak']))"; }
run bert_large
timeout -k 10 300 python bench.py > gpurun_out/table/bert_large_full.json 2> gpurun_out/table/bert_large_full.err
run bert_base_ragged --config bert_base --ragged
run t5_base_ragged --config t5_base --ragged
run electra_large --config electra_large
run electra_large_f32 --config electra_large --dtype f32
run bert_large_from_ids --from-ids
