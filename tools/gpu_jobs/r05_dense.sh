#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05dense; mkdir -p $O
cd $R
python tools/train_dense_time.py 2>$O/err.txt | tee $O/time.jsonl; tail -3 $O/err.txt
for v in $(ls mv2d_amd/lib/variants/ 2>/dev/null | grep "^libda"); do MV2D_HIP_LIB=$R/mv2d_amd/lib/variants/$v python tools/train_dense_time.py 2>/dev/null | tee -a $O/time.jsonl; done
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "dense_block or denoising" 2>&1 | grep -E "passed|failed"
