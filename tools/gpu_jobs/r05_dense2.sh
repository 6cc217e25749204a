#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python tools/train_dense_time.py --p 0 2>/dev/null
python tools/train_dense_time.py --p 0.1 --n 400 --nk 4000 2>/dev/null
