#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "xattn_fused" 2>&1 | tail -3
python bench.py --brief --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', d['value'], 'decoder ms/launch', d['decoder_ms_per_launch'], d.get('index_mismatches'))"
HEAD=6 tools/prof_cmd.sh r05g/prof_exact python tools/run_engine.py --batch 16 --steps 20
