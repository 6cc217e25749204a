#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
for w in cfg3_t cfg5_t cfg2_s_nc6; do timeout 600 python tools/microbench_cluster_order.py $w 2>&1 | grep -v Warn | tail -9; done | tee $O/cluster_order.txt
