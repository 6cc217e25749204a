#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t5; mkdir -p $O
timeout 300 python $R/tools/train_fused_unit.py > $O/unit_cfg2s.json 2>$O/err.txt
tail -5 $O/err.txt
timeout 300 python $R/tools/train_fused_ab.py > $O/ab_cfg2s.json 2>$O/err2.txt
tail -3 $O/err2.txt; cat $O/ab_cfg2s.json
python $R/tools/train_prof_step.py 2>/dev/null
python $R/tools/prof_train_host.py > $O/host.txt 2>&1
