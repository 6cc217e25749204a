#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05flaky; mkdir -p $O
cd $R
for i in 1 2 3; do
  timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee -a $O/runs.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tee -a $O/runs.txt
