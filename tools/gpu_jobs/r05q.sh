#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_golden.py tests/test_gpu_plugin.py tests/test_gpu_logits.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
python bench.py --brief --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', d['value'], 'decoder', d['decoder_ms_per_launch'], d.get('index_mismatches'))"
for w in "cfg3_t 16" "cfg5_t 4"; do set -- $w; python bench.py --brief --steps 60 --warmup 10 --workload $1 --batch $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], (d.get('index_mismatches') or {}).get('index_exact'))"; done
