#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t17; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o s -- python $R/tools/train_prof_step.py --iters 6 > $O/run.json 2>/dev/null
DB=$(find $O/tr -name "s_results.db" | head -1)
python $R/tools/rocpd_timeline.py $DB --last-ms 13 > $O/timeline.txt 2>&1
rm -rf $O/tr
tail -1 $O/timeline.txt; cat $O/run.json
