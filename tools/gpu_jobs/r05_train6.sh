#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t6; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -8 > $O/pytest_train.txt
cat $O/pytest_train.txt
timeout 300 python $R/tools/train_fused_ab.py 2>/dev/null | tee $O/ab_cfg2s.json
python $R/tools/train_prof_step.py 2>/dev/null | tee $O/step.json
python $R/tools/bench_train.py 2>/dev/null | tee $O/train_step_cfg2s.json
