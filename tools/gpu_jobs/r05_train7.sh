#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t7; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -q -k "dense_block" 2>&1 > $O/pytest_train.txt
grep -n "^E \|passed\|failed" $O/pytest_train.txt | cut -c1-600
