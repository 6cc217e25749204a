#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pe_fused_x3" 2>&1 | tail -3
python tools/pe_time.py 250000 1 2>&1 | grep rows
for v in 1 2 5; do MV2D_HIP_LIB=mv2d_amd/lib/variants/libpb$v.so timeout 120 python tools/pe_time.py 250000 1 2>&1 | grep "x3b\|fault"; done
