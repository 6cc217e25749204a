#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t14; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 > $O/pytest_train.txt
grep -n "^E \|passed\|failed" $O/pytest_train.txt | cut -c1-400
python $R/tools/bench_train.py 2>/dev/null | tee $O/train_step_cfg2s.json
python $R/tools/train_prof_step.py 2>/dev/null | tee $O/step.json
python $R/tools/prof_train_host.py 2>&1 | grep -E "forward host|backward host"
