#!/bin/bash
# round 5, job b: GPU tests on the flipped default + first bench / profile of the index-exact route
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
timeout 600 python bench.py --steps 100 --no-other-workloads > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
HEAD=40 tools/prof_cmd.sh r05b/prof_exact python tools/run_engine.py --batch 16 --steps 20 > /dev/null 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05b/bench.json') if l.startswith('{')][-1])
print('value', d['value'], d['route'], 'key16', d.get('samples_s_key16_mode_opt_in'), 'ratio', d.get('index_exact_vs_key16_mode'), 'batch1', d.get('samples_s_batch1'))
print('parity', d.get('ranked_index_mismatches_vs_reference'))
print('roofline', {k: d['roofline'].get(k) for k in ('launch_ms', 'frac', 'frac_at_survey_b2', 'frac_incl_own_intermediates', 'achieved')})
print('cpu', d.get('cpu_baseline'), d.get('cpu_baseline_all_cores'))
PY
