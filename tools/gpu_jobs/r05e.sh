#!/bin/bash
# round 5, job e: the one-launch cross attention -- bitwise test, parity, A/B against the three kernels
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "xattn_fused" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -5
for f in 0 1; do python bench.py --brief --steps 100 --warmup 10 --fuse-xattn $f 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fuse_xattn=$f', d['value'], 'decoder ms/launch', d['decoder_ms_per_launch'], d.get('index_mismatches'))"; done
HEAD=14 tools/prof_cmd.sh r05e/prof_exact python tools/run_engine.py --batch 16 --steps 20
