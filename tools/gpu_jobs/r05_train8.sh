#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t8; mkdir -p $O
timeout 300 python $R/tools/train_fused_unit.py --problem nc6_s > $O/unit_nc6s.json 2>$O/err.txt
tail -3 $O/err.txt
timeout 300 python $R/tools/train_fused_unit.py --problem cfg1_s > $O/unit_cfg1s.json 2>$O/err.txt
