timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -W ignore -x 2>&1 | tail -3
for inf in 1 4; do for w in cfg2_s cfg5_t; do
timeout 120 python bench.py --steps 150 --inflight $inf --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $inf $w', d['value'], d['decoder_ms_per_iter'])"
done; done
