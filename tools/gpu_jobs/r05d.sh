#!/bin/bash
# round 5, job d: pe_x3 feature-row touch variants (per-phase stamps), fast frustum rows test, engine timing
O=gpurun_out/r05d; mkdir -p $O
for t in 0 1 2; do MV2D_HIP_LIB=mv2d_amd/lib/variants/libpxtouch$t.so python tools/px_trace.py 2>&1 | grep -v Warn > $O/px_trace_touch$t.txt; echo touch $t; cat $O/px_trace_touch$t.txt | tr '\n' ';'; echo; done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -x -s -k "frustum or exact_mode" 2>&1 | grep -E "pe_frustum|index parity|passed|failed|Error" | cut -c1-250
HEAD=12 tools/prof_cmd.sh r05d/prof_exact python tools/run_engine.py --batch 16 --steps 20
