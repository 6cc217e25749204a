#!/bin/bash
# round 5, job n: FFN row tiles per block at 4800 rows (tail of the 600-block launch), spread of small fused-attention launches (batch 1)
O=gpurun_out/r05n; mkdir -p $O
for v in "" mv2d_amd/lib/variants/libffn_r2m.so mv2d_amd/lib/variants/libffn_r3m.so; do
  for M in 4800 2400 320; do echo "lib=${v:-default} rows=$M"; MV2D_HIP_LIB=$v python tools/microbench_ffn.py $M 2>/dev/null | grep "8 slices"; done
done | tee $O/ffn_rtb.txt
python bench.py --brief --steps 100 --warmup 10 --batch 1 --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch 1 x 4 streams:', d['value'], 'decoder ms/launch', d['decoder_ms_per_launch'], 'fused', d['roofline']['launch_ms'])"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "xattn_fused" 2>&1 | tail -2
