#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t18; mkdir -p $O
timeout 600 python $R/tools/train_fused_unit.py --problem cfg5_t > $O/unit_cfg5t.json 2>$O/err.txt; tail -3 $O/err.txt
timeout 600 python $R/tools/bench_train.py --problem cfg5_t --iters 5 2>$O/err2.txt | tee $O/train_step_cfg5t.json; tail -3 $O/err2.txt
MV2D_TD_SERIAL=1 timeout 600 python -m pytest $R/tests/test_gpu_train.py -x -q -k "issued_from_c or denoising or forward_train" 2>&1 | tail -2
