#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o s -- python $R/tools/train_decoder_time.py --iters 2 > $O/run.json 2>/dev/null
DB=$(find $O/tr -name "s_results.db" | head -1)
python $R/tools/rocpd_timeline.py $DB --last-ms 14 > $O/timeline.txt 2>&1
rm -rf $O/tr
tail -2 $O/timeline.txt; cat $O/run.json
