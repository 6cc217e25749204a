#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_train.py --problem cfg3_t 2>/dev/null | tee $O/train_step_cfg3t.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/tools/train_prof_step.py --problem cfg3_t > $O/step_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find $O/stats -name "s_results.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/stats
cat $O/step_under_rocprof.json
