#!/bin/bash
# rocprofv3 kernel-trace stats of one bench command: tools/prof_kernels.sh <tag> <bench args...>
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
python tools/rocpd_stats.py $OUT/trace_results.db -100 > $OUT/kernel_stats.txt
grep '"metric"' $OUT/trace.log > $OUT/bench_under_trace.json
cat $OUT/kernel_stats.txt
