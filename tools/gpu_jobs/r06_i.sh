#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "xattn_fused" 2>&1 | tail -2
MV2D_HIP_LIB=mv2d_amd/lib/variants/libxftrace.so python tools/xf_trace.py cfg2_s 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('cfg2_s', d['value'], 'xattn idle ms', d['roofline'].get('launch_ms_idle_gpu'))"
timeout 600 python bench.py --workload cfg2_s_nc6 --steps 20 --warmup 5 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('cfg2_s_nc6', d['value'], 'xattn idle ms', d['roofline'].get('launch_ms_idle_gpu'))"
