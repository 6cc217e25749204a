#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_golden.py tests/test_gpu_plugin.py -m gpu -q -x 2>&1 | tail -3
python bench.py --brief --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', d['value'], d['config']['workload'][:150], d['stage_ms'], d.get('index_mismatches'))"
python bench.py --brief --steps 60 --warmup 10 --workload cfg2_s_nc6 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nc6 value', d['value'], d.get('index_mismatches'))"
