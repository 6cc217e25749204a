#!/bin/bash
# round 5, job f: fused kernel bitwise test; sweep of samples per launch x streams on the index-exact route
O=gpurun_out/r05f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "xattn_fused or xattn_tile" 2>&1 | tail -3
for cfg in "12 4" "16 4" "20 4" "24 4" "16 3" "24 3" "32 3" "16 5"; do set -- $cfg
  python bench.py --brief --steps 60 --warmup 10 --prime 60 --batch $1 --inflight $2 --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch $1 x streams $2:', d['value'], 'samples/s')"
done | tee $O/sweep.txt
