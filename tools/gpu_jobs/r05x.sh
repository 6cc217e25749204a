#!/bin/bash
O=gpurun_out/r05x; mkdir -p $O
for cfg in "16 4" "24 4" "32 4" "16 3" "24 3" "32 3" "48 2" "12 6" "16 6" "8 8"; do set -- $cfg
  python bench.py --brief --steps 60 --warmup 10 --prime 60 --batch $1 --inflight $2 --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch $1 x streams $2:', d['value'], 'samples/s')"
done | tee $O/sweep.txt
