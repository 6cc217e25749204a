# the two bench JSONs of tools/r06_evidence.sh step 1 again, now that profiles/ holds the kernel trace and the counters of THIS library
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python bench.py > $O/default_bench_cfg2s.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape_bench_cfg2s.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ('default_bench_cfg2s', 'driver_shape_bench_cfg2s'):
    d = json.loads([l for l in open('gpurun_out/r06/%s.json' % f) if l.startswith('{')][-1])
    print(f, d['value'], 'timed s', d.get('timed_seconds'), 'key16', d.get('samples_s_key16_mode_opt_in'), 'batch1', d.get('samples_s_batch1'), 'nchw', d.get('samples_s_nchw_input'))
    print('  roofline', {k: d['roofline'].get(k) for k in ('launch_ms', 'launch_ms_idle_gpu', 'launch_ms_rocprof_committed', 'bytes_per_launch', 'frac', 'frac_at_b4', 'frac_at_survey_b2', 'traffic', 'traffic_stale')})
    print('  other', {k: (v.get('value'), v['roofline'].get('frac'), v['roofline'].get('launch_ms')) for k, v in (d.get('other_workloads') or {}).items()})
PY
