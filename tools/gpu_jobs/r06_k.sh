#!/bin/bash
# streams x samples per launch sweep of the headline loop (index-exact route, channels_last maps)
for cfg in "4 16" "5 16" "6 16" "8 16" "4 24" "4 32" "3 32" "6 8"; do set -- $cfg
  timeout 600 python bench.py --inflight $1 --batch $2 --steps 20 --warmup 5 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('streams $1 x batch $2:', d['value'], 'samples/s, timed', d.get('timed_seconds'), 's')"
done
