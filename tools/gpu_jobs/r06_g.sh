#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline --no-collective-leg > $O/a.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --brief --nchw-input > $O/b.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 120 --warmup 5 --brief --rounds 1 > $O/c.json 2>> $O/bench.err
python - <<'PY'
import json
for f in 'abc':
    d = json.loads([l for l in open('gpurun_out/r06g/%s.json' % f) if l.startswith('{')][-1])
    print(f, 'value', d['value'], 'ms/step', d['ms_per_step'], 'timed s', d.get('timed_seconds'), d['config'].get('feature_map_memory'), 'rounds', d['config'].get('launch_sequences_per_stream_and_step'))
    if f == 'a':
        print('  nchw leg', d.get('samples_s_nchw_input'), 'fixed', d.get('samples_s_fixed_inputs'), 'key16', d.get('samples_s_key16_mode_opt_in'), 'batch1', d.get('samples_s_batch1'), 'lat', d.get('latency_ms_single_stream'))
        print('  roofline', {k: d['roofline'].get(k) for k in ('launch_ms', 'launch_ms_idle_gpu', 'frac', 'frac_at_survey_b2')})
PY
