#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t1; mkdir -p $O
python $R/tools/train_gemm_shapes.py > $O/shapes_cfg2s.jsonl 2>$O/err.txt
tail -3 $O/err.txt; wc -l $O/shapes_cfg2s.jsonl
