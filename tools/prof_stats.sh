#!/bin/bash
# usage: tools/prof_stats.sh <outdir-under-gpurun_out> <bench args...>   (kernel-trace stats of one bench command)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$OUT/stats -o s -- python $R/bench.py --steps 100 --no-cpu-baseline --no-other-workloads "$@" > $R/gpurun_out/$OUT/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/$OUT/stats -name "s_results.db" | head -1) > $R/gpurun_out/$OUT/kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/$OUT/stats
head -${HEAD:-14} $R/gpurun_out/$OUT/kernel_stats.txt
