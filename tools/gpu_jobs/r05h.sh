#!/bin/bash
# round 5, job h: T-path workloads on the index-exact route (and the key16 mode beside them), per-kernel table of cfg3_t / cfg5_t
O=gpurun_out/r05h; mkdir -p $O
for w in "cfg3_t 16" "cfg5_t 4" "cfg2_s_nc6 16"; do set -- $w
  for m in "" "--key16"; do python bench.py --brief --steps 60 --warmup 10 --workload $1 --batch $2 $m 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 $m', d['value'], 'decoder ms/launch', d['decoder_ms_per_launch'], 'tile', d['roofline']['launch_ms'], d['roofline']['frac'], (d.get('index_mismatches') or {}))"; done
done | tee $O/t_path.txt
HEAD=16 tools/prof_cmd.sh r05h/prof_cfg3t python tools/run_engine.py --workload cfg3_t --batch 16 --steps 20
HEAD=10 tools/prof_cmd.sh r05h/prof_cfg5t python tools/run_engine.py --workload cfg5_t --batch 4 --steps 20
