#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -3
python bench.py --brief --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', d['value'], 'decoder ms/launch', d['decoder_ms_per_launch'], d.get('index_mismatches'))"
