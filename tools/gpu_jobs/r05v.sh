#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -k "xattn_fused or query_order or golden or bitwise" 2>&1 | tail -2
for w in "cfg2_s 16" "cfg2_s_nc6 16"; do set -- $w; python bench.py --brief --steps 100 --warmup 10 --workload $1 --batch $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'fused us', d['roofline']['launch_ms'], d['roofline']['frac'], (d.get('index_mismatches') or {}).get('index_exact'))"; done
python bench.py --brief --steps 100 --warmup 10 --batch 1 --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch 1 x 4 streams:', d['value'], 'fused', d['roofline']['launch_ms'])"
( time python bench.py --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1 ) 2>&1 | grep real
