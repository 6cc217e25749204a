#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t16; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 > $O/pytest_train.txt
grep -n "^E \|passed\|failed" $O/pytest_train.txt | cut -c1-400
python $R/tools/bench_train.py --problem cfg3_t 2>/dev/null | tee $O/train_step_cfg3t.json
python $R/tools/bench_train.py 2>/dev/null | tee $O/train_step_cfg2s.json
