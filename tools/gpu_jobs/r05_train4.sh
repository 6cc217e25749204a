#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/train_prof_step.py > $O/step_plain.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/tools/train_prof_step.py > $O/step_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find $O/stats -name "s_results.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/stats
python $R/tools/prof_train_host.py > $O/host.txt 2>&1
cat $O/step_plain.json; head -3 $O/kernel_stats.txt | cut -c1-150
