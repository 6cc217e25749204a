#!/bin/bash
# round 6, first call: the shared-tile cross attention (csrc/xattn_group.hip): unit test, T-path goldens, per-kernel times old vs new
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "xattn_group" 2>&1 | tail -15 > $O/unit.txt
cat $O/unit.txt
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_logits.py tests/test_gpu_engine.py -q -x -s 2>&1 | grep -E "index parity|passed|failed|Error|error|assert" | tail -30 > $O/golden.txt
cat $O/golden.txt
for w in "cfg3_t --batch 16" "cfg5_t --batch 4"; do
  set -- $w; n=$1
  timeout 300 python tools/run_engine.py --workload "$@" --steps 20 2>&1 | tail -1
  timeout 300 python tools/run_engine.py --workload "$@" --steps 20 --group 0 2>&1 | tail -1
  HEAD=14 tools/prof_cmd.sh r06a/prof_$n python tools/run_engine.py --workload "$@" --steps 10 2>&1 | tail -16
done
