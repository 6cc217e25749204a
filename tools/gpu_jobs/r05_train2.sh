#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t2; mkdir -p $O
timeout 300 python $R/tools/train_fused_ab.py > $O/ab_cfg2s.json 2>$O/err.txt
tail -5 $O/err.txt; cat $O/ab_cfg2s.json
