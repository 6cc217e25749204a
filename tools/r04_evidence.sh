#!/bin/bash
# Round-4 evidence, one call on the GPU box: bash tools/r04_evidence.sh   -> gpurun_out/r04/* (copied to profiles/r04_* afterwards)
O=gpurun_out/r04
mkdir -p $O
# 0. fp16 MFMA keeps subnormal inputs (the hi + lo key rows rely on it)
timeout 60 mv2d_amd/lib/f16_mfma_probe > $O/f16_mfma_probe.txt 2>&1
# 1. the default bench line (CPU baseline leg, index-exact leg, one-rank RCCL leg, mismatch counts, other workloads) and the driver's call shape
timeout 900 python bench.py --steps 100 > $O/default_bench_cfg2s.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape_bench_cfg2s.json 2>> $O/bench.err
# 2. rocprofv3 kernel summary of the same default command (extra legs off) + PMC passes of the eager single-stream bench per workload
HEAD=5 tools/prof_stats.sh r04/stats_default --no-extra-legs --no-parity-leg --no-collective-leg > /dev/null 2>&1
mv $O/stats_default/kernel_stats.txt $O/default_bench_cfg2s_kernel_stats.txt; mv $O/stats_default/bench_under_rocprof.json $O/default_bench_cfg2s_under_rocprof.json; rmdir $O/stats_default
L2=1 STATS=0 bash tools/pmc_bench.sh r04/pmc_cfg2s > /dev/null 2>&1
L2=1 STATS=0 bash tools/pmc_bench.sh r04/pmc_cfg2s_nc6 --workload cfg2_s_nc6 > /dev/null 2>&1
L2=1 STATS=0 bash tools/pmc_bench.sh r04/pmc_cfg3t --workload cfg3_t --batch 16 > /dev/null 2>&1
L2=1 STATS=0 bash tools/pmc_bench.sh r04/pmc_cfg5t --workload cfg5_t --batch 4 > /dev/null 2>&1
# 3. the one-rank RCCL leg under the kernel trace: the all-gather kernel beside the frame kernels
HEAD=40 tools/prof_cmd.sh r04/prof_rccl python bench.py --brief --force-collective --steps 60 --warmup 10 --no-parity-leg > /dev/null 2>&1
timeout 300 python bench.py --brief --force-collective --steps 200 --warmup 10 --no-parity-leg > $O/bench_cfg2s_one_rank_rccl.json 2>> $O/bench.err
# 4. per-kernel tables of one stream replaying 16-sample frames: default / index-exact route, S and T path; one sample per launch
for w in "default_cfg2s --batch 16" "exact_cfg2s --exact --batch 16" "default_cfg3t --workload cfg3_t --batch 16" "exact_cfg3t --exact --workload cfg3_t --batch 16" \
         "default_cfg5t --workload cfg5_t --batch 4" "default_cfg2s_batch1 --batch 1"; do
  set -- $w; n=$1; shift
  HEAD=60 tools/prof_cmd.sh r04/prof_$n python tools/run_engine.py "$@" --steps 20 > /dev/null 2>&1
  mv $O/prof_$n/kernel_stats.txt $O/engine_${n}_kernel_stats.txt; rm -rf $O/prof_$n
done
# 5. which key-side rounding costs how many ranks
python tools/ablate_exact.py 2>/dev/null > $O/ablate_exact.txt
# 6. training step (autograd route on the HIP kernels)
python tools/bench_train.py 2>/dev/null | tail -n 1 > $O/train_step_cfg2s.json
python tools/bench_train.py --problem cfg3_t 2>/dev/null | tail -n 1 > $O/train_step_cfg3t.json
# 7. the GPU test suite with the parity prints
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "index parity|logit parity|float parity|tie gap|keep rate|passed|failed|^(micro|cfg|nc6)[a-z0-9_]* R " > $O/gpu_tests_parity_lines.txt
tail -2 $O/gpu_tests_parity_lines.txt; tail -c 300 $O/bench.err
python - <<'PY'
import json
for f in ('default_bench_cfg2s', 'driver_shape_bench_cfg2s'):
    d = json.loads([l for l in open('gpurun_out/r04/%s.json' % f) if l.startswith('{')][-1])
    print(f, d['value'], 'coll', d.get('samples_s_with_collective'), 'exact', d.get('samples_s_index_exact'), d.get('index_exact_vs_default'), 'batch1', d.get('samples_s_batch1'),
          'batch8', d.get('samples_s_batch8'))
    print('  parity', d.get('ranked_index_mismatches_vs_reference'))
    print('  other', {k: (v.get('value'), v.get('index_exact_vs_default')) for k, v in (d.get('other_workloads') or {}).items()})
PY
