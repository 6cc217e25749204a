#!/bin/bash
# Round-5 evidence, one call on the GPU box: bash tools/r05_evidence.sh   -> gpurun_out/r05/* (tools/copy_evidence.sh copies it to profiles/r05_*)
O=gpurun_out/r05
mkdir -p $O
# 1. the default bench line (index-exact route; CPU baseline legs, key16-mode leg, one-rank RCCL leg, mismatch counts, other workloads) and the driver's call shape
timeout 1200 python bench.py --steps 100 > $O/default_bench_cfg2s.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape_bench_cfg2s.json 2>> $O/bench.err
# 2. rocprofv3 kernel summary of the same default command (extra legs off) + PMC passes of the eager single-stream bench per workload
HEAD=5 tools/prof_stats.sh r05/stats_default --no-extra-legs --no-parity-leg --no-collective-leg > /dev/null 2>&1
mv $O/stats_default/kernel_stats.txt $O/default_bench_cfg2s_kernel_stats.txt; mv $O/stats_default/bench_under_rocprof.json $O/default_bench_cfg2s_under_rocprof.json; rmdir $O/stats_default
L2=1 STATS=0 bash tools/pmc_bench.sh r05/pmc_cfg2s > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r05/pmc_cfg2s_nc6 --workload cfg2_s_nc6 > /dev/null 2>&1
L2=1 STATS=0 bash tools/pmc_bench.sh r05/pmc_cfg3t --workload cfg3_t --batch 16 > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r05/pmc_cfg5t --workload cfg5_t --batch 4 > /dev/null 2>&1
STATS=0 bash tools/pmc_bench.sh r05/pmc_cfg2s_key16 --key16 > /dev/null 2>&1
# 3. the one-rank RCCL leg under the kernel trace
HEAD=40 tools/prof_cmd.sh r05/prof_rccl python bench.py --brief --force-collective --steps 60 --warmup 10 --no-parity-leg > /dev/null 2>&1
timeout 300 python bench.py --brief --force-collective --steps 200 --warmup 10 --no-parity-leg > $O/bench_cfg2s_one_rank_rccl.json 2>> $O/bench.err
# 4. per-kernel tables of one stream replaying 16-sample frames: index-exact route (S, T), the opt-in key16 mode, one sample per launch
for w in "exact_cfg2s --batch 16" "key16_cfg2s --key16 --batch 16" "exact_cfg3t --workload cfg3_t --batch 16" "exact_cfg5t --workload cfg5_t --batch 4" \
         "exact_cfg2s_nc6 --workload cfg2_s_nc6 --batch 16" "exact_cfg2s_batch1 --batch 1"; do
  set -- $w; n=$1; shift
  HEAD=60 tools/prof_cmd.sh r05/prof_$n python tools/run_engine.py "$@" --steps 20 > /dev/null 2>&1
  mv $O/prof_$n/kernel_stats.txt $O/engine_${n}_kernel_stats.txt; rm -rf $O/prof_$n
done
# 5. which key-side rounding costs how many ranks (query side in fp16 pairs)
python tools/ablate_exact.py 2>/dev/null > $O/ablate_exact.txt
# 6. training step (autograd route on the HIP kernels)
python tools/bench_train.py 2>/dev/null | tail -n 1 > $O/train_step_cfg2s.json
python tools/bench_train.py --problem cfg3_t 2>/dev/null | tail -n 1 > $O/train_step_cfg3t.json
#    ... the same step with the per-operator graph of rounds 3-4, the decoder + branches alone (side streams on / off / per-operator), the host profile
MV2D_TRAIN_FUSED=0 python tools/bench_train.py 2>/dev/null | tail -n 1 > $O/train_step_cfg2s_operator_graph.json
(python tools/train_decoder_time.py; MV2D_TD_SERIAL=1 python tools/train_decoder_time.py; MV2D_TRAIN_FUSED=0 python tools/train_decoder_time.py) 2>/dev/null | grep '^{' > $O/train_decoder_time.jsonl
python tools/prof_train_host.py 2>&1 | grep -v Warning > $O/train_step_cfg2s_host_profile.txt
R_=$PWD; (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats -d $R_/$O/train_stats -o s -- python $R_/tools/train_prof_step.py > $R_/$O/train_step_cfg2s_under_rocprof.json 2>/dev/null)
python tools/rocpd_stats.py $(find $O/train_stats -name "s_results.db" | head -1) > $O/train_step_cfg2s_kernel_stats.txt 2>&1; rm -rf $O/train_stats
# 7. per-phase stamps of the PE kernel
python tools/px_trace.py > /dev/null 2>&1 || true
# 8. the GPU test suite with the parity prints
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "index parity|logit parity|float parity|tie gap|keep rate|pe_frustum|passed|failed|skipped|^(micro|cfg|nc6)[a-z0-9_]* R " > $O/gpu_tests_parity_lines.txt
tail -2 $O/gpu_tests_parity_lines.txt; tail -c 300 $O/bench.err
python - <<'PY'
import json
for f in ('default_bench_cfg2s', 'driver_shape_bench_cfg2s'):
    d = json.loads([l for l in open('gpurun_out/r05/%s.json' % f) if l.startswith('{')][-1])
    print(f, d['value'], d['route'], 'coll', d.get('samples_s_with_collective'), 'key16', d.get('samples_s_key16_mode_opt_in'), d.get('index_exact_vs_key16_mode'), 'batch1', d.get('samples_s_batch1'),
          'batch8', d.get('samples_s_batch8'))
    print('  parity', d.get('ranked_index_mismatches_vs_reference'))
    print('  roofline', {k: d['roofline'].get(k) for k in ('kernel', 'launch_ms', 'frac', 'frac_at_survey_b2', 'traffic')})
    print('  cpu', d.get('cpu_baseline'), d.get('cpu_baseline_all_cores'))
    print('  other', {k: (v.get('value'), v.get('index_exact_vs_key16_mode')) for k, v in (d.get('other_workloads') or {}).items()})
PY
