#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "query_order or bitwise or route_options" 2>&1 | tail -3
for w in "cfg2_s 16" "cfg2_s_nc6 16"; do set -- $w; python bench.py --brief --steps 100 --warmup 10 --workload $1 --batch $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'fused us', d['roofline']['launch_ms'], d['roofline']['frac'], (d.get('index_mismatches') or {}).get('index_exact'))"; done
