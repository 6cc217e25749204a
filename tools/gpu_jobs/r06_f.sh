#!/bin/bash
# round 6: the whole GPU suite + the bench line in the driver's call shape
O=gpurun_out/r06f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06f/driver_shape.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'timed s', d.get('timed_seconds'), 'route', d['route'], 'long_run', d.get('long_run'))
print('nchw leg', d.get('samples_s_nchw_input'), 'key16', d.get('samples_s_key16_mode_opt_in'), 'batch1', d.get('samples_s_batch1'), 'lat', d.get('latency_ms_single_stream'))
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'launch_ms', 'launch_ms_idle_gpu', 'frac', 'frac_at_survey_b2', 'traffic')})
print('cpu', d.get('cpu_baseline'))
print('cpu all', d.get('cpu_baseline_all_cores'))
print('parity', d.get('ranked_index_mismatches_vs_reference'))
print('other', {k: (v.get('value')) for k, v in (d.get('other_workloads') or {}).items()})
print('coll', d.get('collective_leg'))
PY
