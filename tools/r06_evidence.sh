#!/bin/bash
# Round-6 evidence, one call on the GPU box: bash tools/r06_evidence.sh   -> gpurun_out/r06/* (tools/copy_evidence_r06.sh copies it to profiles/r06_*)
O=gpurun_out/r06
mkdir -p $O
# 1. the default bench line (index-exact route; CPU baseline legs, key16-mode leg, one-rank RCCL leg, mismatch counts, other workloads) and the driver's call shape
timeout 1500 python bench.py > $O/default_bench_cfg2s.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape_bench_cfg2s.json 2>> $O/bench.err
# 1b. a sustained run: 640 steps of 384 frames (~29 s of timed steps) -- clocks and throughput under a long load
timeout 900 python bench.py --steps 640 --warmup 20 --brief --no-parity-leg 2>> $O/bench.err | tail -n 1 > $O/soak_bench_cfg2s.json
# 2. rocprofv3 kernel summary of the default command (extra legs off) + PMC passes of the eager single-stream bench per workload
HEAD=5 tools/prof_stats.sh r06/stats_default --steps 40 --no-extra-legs --no-parity-leg --no-collective-leg > /dev/null 2>&1
mv $O/stats_default/kernel_stats.txt $O/default_bench_cfg2s_kernel_stats.txt; mv $O/stats_default/bench_under_rocprof.json $O/default_bench_cfg2s_under_rocprof.json; rmdir $O/stats_default
L2=1 STATS=0 bash tools/pmc_bench.sh r06/pmc_cfg2s --rounds 1 > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r06/pmc_cfg2s_nc6 --rounds 1 --workload cfg2_s_nc6 > /dev/null 2>&1
L2=1 STATS=0 bash tools/pmc_bench.sh r06/pmc_cfg3t --rounds 1 --workload cfg3_t --batch 16 > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r06/pmc_cfg5t --rounds 1 --workload cfg5_t --batch 4 > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r06/pmc_cfg2s_key16 --rounds 1 --key16 > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r06/pmc_cfg2s_nchw --rounds 1 --nchw-input > /dev/null 2>&1
# 3. the one-rank RCCL leg under the kernel trace
HEAD=40 tools/prof_cmd.sh r06/prof_rccl python bench.py --brief --force-collective --steps 20 --warmup 5 --no-parity-leg > /dev/null 2>&1
timeout 300 python bench.py --brief --force-collective --steps 40 --warmup 5 --no-parity-leg > $O/bench_cfg2s_one_rank_rccl.json 2>> $O/bench.err
# 4. per-kernel tables of one stream replaying 16-sample frames: index-exact route (S, T), the opt-in key16 mode, one sample per launch, the two opt-in round-6 kernels
for w in "exact_cfg2s --batch 16" "key16_cfg2s --key16 --batch 16" "exact_cfg3t --workload cfg3_t --batch 16" "exact_cfg5t --workload cfg5_t --batch 4" \
         "exact_cfg2s_nc6 --workload cfg2_s_nc6 --batch 16" "exact_cfg2s_batch1 --batch 1" "optin_group_xattn_cfg3t --workload cfg3_t --batch 16 --group 1" \
         "optin_pe_rows_in_waves_cfg3t --workload cfg3_t --batch 16 --pe-v2"; do
  set -- $w; n=$1; shift
  HEAD=60 tools/prof_cmd.sh r06/prof_$n python tools/run_engine.py "$@" --steps 20 > /dev/null 2>&1
  mv $O/prof_$n/kernel_stats.txt $O/engine_${n}_kernel_stats.txt; rm -rf $O/prof_$n
done
# 5. the round-6 experiments: launch order / shared-tile cross attention, the two shapes of the PE kernel, queries per block of the one-launch attention
for w in cfg3_t cfg5_t cfg2_s_nc6; do timeout 600 python tools/microbench_cluster_order.py $w 2>&1 | grep -v Warn | tail -9; done > $O/cluster_order_and_shared_tiles.txt
(python tools/pe_time.py 250000 1; python tools/pe_time.py 100000 0) 2>&1 | grep rows > $O/pe_kernel_shapes.txt
for qb in 8 4; do MV2D_XF_QB=$qb timeout 600 python bench.py --steps 20 --warmup 5 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('xattn_fused queries per block = $qb: cfg2_s', d['value'], 'samples/s, kernel', d['roofline'].get('launch_ms_idle_gpu'), 'ms per launch (idle GPU)')"; done > $O/xattn_fused_queries_per_block.txt
bash tools/pmc_group.sh cfg3_t 16 > /dev/null 2>&1; cp gpurun_out/pmcgroup_cfg3_t/summary.txt $O/pmc_xattn_group_cfg3t.txt
# 6. which key-side rounding costs how many ranks (query side in fp16 pairs)
python tools/ablate_exact.py 2>/dev/null > $O/ablate_exact.txt
# 7. training step (continuity with round 5; untouched this round)
python tools/bench_train.py 2>/dev/null | tail -n 1 > $O/train_step_cfg2s.json
python tools/bench_train.py --problem cfg3_t 2>/dev/null | tail -n 1 > $O/train_step_cfg3t.json
# 8. the GPU test suite with the parity prints
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "index parity|logit parity|float parity|tie gap|keep rate|pe_frustum|xattn_group|passed|failed|skipped|^(micro|cfg|nc6)[a-z0-9_]* R " > $O/gpu_tests_parity_lines.txt
tail -2 $O/gpu_tests_parity_lines.txt; tail -c 300 $O/bench.err
python - <<'PY'
import json
for f in ('default_bench_cfg2s', 'driver_shape_bench_cfg2s'):
    d = json.loads([l for l in open('gpurun_out/r06/%s.json' % f) if l.startswith('{')][-1])
    print(f, d['value'], d['route'], 'timed s', d.get('timed_seconds'), 'coll', d.get('samples_s_with_collective'), 'key16', d.get('samples_s_key16_mode_opt_in'), d.get('index_exact_vs_key16_mode'), 'batch1', d.get('samples_s_batch1'),
          'batch8', d.get('samples_s_batch8'), 'nchw', d.get('samples_s_nchw_input'))
    print('  parity', {k: v for k, v in (d.get('ranked_index_mismatches_vs_reference') or {}).items() if not k.startswith('reference')})
    print('  roofline', {k: d['roofline'].get(k) for k in ('kernel', 'launch_ms', 'launch_ms_idle_gpu', 'launch_ms_rocprof_committed', 'frac', 'frac_at_survey_b2', 'traffic')})
    print('  cpu', d.get('cpu_baseline'), d.get('cpu_baseline_all_cores'))
    print('  other', {k: (v.get('value'), v.get('index_exact_vs_key16_mode')) for k, v in (d.get('other_workloads') or {}).items()})
PY
