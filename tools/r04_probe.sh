#!/bin/bash
# Round-4 first GPU call: bash tools/r04_probe.sh  -> gpurun_out/r04a/*
O=gpurun_out/r04a
mkdir -p $O
# 1. fp16 MFMA keeps subnormal inputs? (the fp16 hi + lo key rows rely on it)
timeout 60 mv2d_amd/lib/f16_mfma_probe > $O/f16_mfma_probe.txt 2>&1; cat $O/f16_mfma_probe.txt | tail -3
# 2. RCCL on one rank: the bench step with the process group initialised and the per-step all-gather in it
timeout 300 python bench.py --brief --force-collective --steps 200 --warmup 10 --no-parity-leg > $O/bench_cfg2s_one_rank_rccl.json 2> $O/coll.err
tail -c 600 $O/coll.err
python - <<'EOF'
import json
d = json.loads([l for l in open('gpurun_out/r04a/bench_cfg2s_one_rank_rccl.json') if l.startswith('{')][-1])
print('one-rank RCCL leg:', d['value'], d.get('collective_check'))
EOF
HEAD=12 tools/prof_cmd.sh r04a/prof_rccl python bench.py --brief --force-collective --steps 60 --warmup 10 --no-parity-leg > /dev/null 2>&1
grep -i -n "nccl\|rccl\|AllGather\|xattn_tile" $O/prof_rccl/kernel_stats.txt | head -8
# 3. T-path counters: HBM fetch / write + L2 hit / miss + TCP->TCC requests per kernel
L2=1 STATS=0 bash tools/pmc_bench.sh r04a/pmc_cfg3t --workload cfg3_t --batch 16 > /dev/null 2>&1
L2=1 STATS=0 bash tools/pmc_bench.sh r04a/pmc_cfg5t --workload cfg5_t --batch 4 > /dev/null 2>&1
grep -h "xattn_tile" $O/pmc_cfg3t/*.txt $O/pmc_cfg5t/*.txt | cut -c1-20,60-140
# 4. the default bench line of HEAD on this box (with the one-rank RCCL leg as a sub-process)
timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/default_bench_cfg2s.json 2> $O/bench.err
python - <<'EOF'
import json
d = json.loads([l for l in open('gpurun_out/r04a/default_bench_cfg2s.json') if l.startswith('{')][-1])
print('default:', d['value'], 'with collective:', d.get('samples_s_with_collective'), d.get('collective_leg'))
print('batch1', d.get('samples_s_batch1'), 'exact', d.get('samples_s_index_exact'), {k: v.get('value') for k, v in (d.get('other_workloads') or {}).items()})
EOF
