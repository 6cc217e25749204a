#!/bin/bash
# usage: tools/prof_cmd.sh <outdir-under-gpurun_out> <command ...>   (rocprofv3 kernel-trace stats of any command run from the repo root)
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$OUT/stats -o s -- "$@" > $R/gpurun_out/$OUT/cmd_stdout.txt 2> $R/gpurun_out/$OUT/cmd_stderr.txt )
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/$OUT/stats -name "s_results.db" | head -1) > $R/gpurun_out/$OUT/kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/$OUT/stats
head -${HEAD:-30} $R/gpurun_out/$OUT/kernel_stats.txt
