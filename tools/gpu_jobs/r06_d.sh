#!/bin/bash
# round 6: second shape of the split-precision PE kernel (csrc/pe_x3b.hip): bitwise test, goldens, times
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pe_fused_x3" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_golden.py -q -x 2>&1 | tail -3
for w in "cfg2_s --batch 16" "cfg3_t --batch 16"; do
  set -- $w; n=$1
  timeout 300 python tools/run_engine.py --workload "$@" --steps 20 2>&1 | tail -1
  timeout 300 python tools/run_engine.py --workload "$@" --steps 20 --pe-v1 2>&1 | tail -1 | sed "s/^/[pe v1] /"
  HEAD=5 tools/prof_cmd.sh r06d/prof_$n python tools/run_engine.py --workload "$@" --steps 10 2>&1 | tail -4
done
