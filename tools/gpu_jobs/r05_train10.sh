#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t10; mkdir -p $O
python $R/tools/train_decoder_time.py 2>/dev/null | tee $O/time.jsonl
MV2D_TD_SERIAL=1 python $R/tools/train_decoder_time.py 2>/dev/null | tee -a $O/time.jsonl
MV2D_TRAIN_FUSED=0 python $R/tools/train_decoder_time.py 2>/dev/null | tee -a $O/time.jsonl
