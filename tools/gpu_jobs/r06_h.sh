#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
MV2D_XF_QB=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "xattn_fused" 2>&1 | tail -2
for qb in 8 4; do
  for w in "cfg2_s 16" "cfg2_s_nc6 16"; do set -- $w
    MV2D_XF_QB=$qb timeout 600 python bench.py --workload $1 --batch $2 --steps 20 --warmup 5 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('QB=$qb', '$1', d['value'], 'xattn idle ms', d['roofline'].get('launch_ms_idle_gpu'), 'decoder ms/launch', d['decoder_ms_per_launch'])"
  done
done
MV2D_XF_QB=4 timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --brief --no-parity-leg --rounds 1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('QB=4 batch1', d['value'])"
MV2D_XF_QB=8 timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --brief --no-parity-leg --rounds 1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('QB=8 batch1', d['value'])"
