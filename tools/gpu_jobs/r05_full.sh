#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05full; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 > $O/pytest_gpu.txt
grep -n "^E \|passed\|failed" $O/pytest_gpu.txt | cut -c1-400
python $R/tools/bench_train.py 2>/dev/null | tee $O/train_step_cfg2s.json
