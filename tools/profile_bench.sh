#!/bin/bash
# Round profile of bench.py on the GPU box: kernel stats + separate HBM-traffic PMC passes.
#   gpurun -- bash tools/profile_bench.sh r01        -> gpurun_out/prof_r01/{stats,fetch,write}
# (each rocprofv3 pass is its own run; --pmc is never combined with other trace domains)
set -e
TAG=${1:-r01}
shift || true
ARGS="$@"     # extra bench.py flags (e.g. --config t5_base --ragged): kernel stats of another config
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $ARGS > $OUT/write.log 2>&1
python $R/tools/profile_post.py $OUT
