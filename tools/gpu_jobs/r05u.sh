#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for w in "cfg2_s 16" "cfg3_t 16"; do set -- $w; python bench.py --brief --steps 100 --warmup 10 --workload $1 --batch $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], (d.get('index_mismatches') or {}).get('index_exact'))"; done
