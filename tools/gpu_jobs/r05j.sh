#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
python tools/microbench_tile_order.py cfg3_t 16 2>&1 | grep -v Warn | tee $O/tile_order_cfg3t.txt
python tools/microbench_tile_order.py cfg5_t 4 2>&1 | grep -v Warn | tee $O/tile_order_cfg5t.txt
