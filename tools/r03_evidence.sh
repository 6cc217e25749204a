#!/bin/bash
# Round-3 evidence, one call on the GPU box: bash tools/r03_evidence.sh   -> gpurun_out/r03/* (copied to profiles/r03_* afterwards)
O=gpurun_out/r03
mkdir -p $O
# 1. the default bench line (CPU baseline leg, exact-mode leg, mismatch counts, other workloads) and the driver's call shape (--steps 20)
timeout 600 python bench.py --steps 100 > $O/default_bench_cfg2s.json 2> $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape_bench_cfg2s.json 2>> $O/bench.err
# 2. rocprofv3 kernel summary of the same default command (extra legs off) + PMC FETCH / WRITE passes of the eager single-stream bench
HEAD=5 tools/prof_stats.sh r03/stats_default --no-extra-legs --no-parity-leg > /dev/null 2>&1
mv $O/stats_default/kernel_stats.txt $O/default_bench_cfg2s_kernel_stats.txt; mv $O/stats_default/bench_under_rocprof.json $O/default_bench_cfg2s_under_rocprof.json; rmdir $O/stats_default
bash tools/pmc_bench.sh r03/pmc_cfg2s > /dev/null 2>&1
# 3. T path: bench lines, per-kernel stats with the ordered per-query kernel, the kernel microbenchmark (query tiles vs per-query, ordered or not)
timeout 300 python bench.py --steps 100 --workload cfg3_t --no-cpu-baseline --no-extra-legs > $O/bench_cfg3t.json 2>> $O/bench.err
timeout 300 python bench.py --steps 100 --workload cfg5_t --batch 4 --no-cpu-baseline --no-extra-legs > $O/bench_cfg5t.json 2>> $O/bench.err
HEAD=40 tools/prof_cmd.sh r03/prof_cfg3t python tools/run_engine.py --workload cfg3_t --steps 20 > /dev/null 2>&1
HEAD=40 tools/prof_cmd.sh r03/prof_cfg5t python tools/run_engine.py --workload cfg5_t --batch 2 --steps 20 > /dev/null 2>&1
timeout 200 python tools/microbench_qtile.py --workload cfg3_t --batch 8 2>/dev/null > $O/microbench_xattn_cfg3t.txt
timeout 200 python tools/microbench_qtile.py --workload cfg5_t --batch 2 2>/dev/null > $O/microbench_xattn_cfg5t.txt
# 4. S path sweep over the RoIs read per query
for nc in 1 2 6; do timeout 200 python bench.py --steps 100 --brief --corr-topk 1 --force-nc $nc > $O/bench_cfg2s_forced_nc$nc.json 2>> $O/bench.err; done
# 5. the index-exact route and the default route on one stream: per-kernel tables; L2 counters of the row kernels
HEAD=40 tools/prof_cmd.sh r03/prof_exact python tools/run_engine.py --exact --steps 20 > /dev/null 2>&1
HEAD=40 tools/prof_cmd.sh r03/prof_default python tools/run_engine.py --steps 20 > /dev/null 2>&1
tools/pmc_engine.sh r03/pmc_engine > /dev/null 2>&1
# 6. training step (autograd route on the HIP kernels)
python tools/bench_train.py 2>/dev/null | tail -n 1 > $O/train_step_cfg2s.json
python tools/bench_train.py --problem cfg3_t 2>/dev/null | tail -n 1 > $O/train_step_cfg3t.json
HEAD=60 tools/prof_cmd.sh r03/prof_train python tools/bench_train.py --iters 5 > /dev/null 2>&1
# 7. the GPU test suite with the parity prints
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "index parity|logit parity|query tiles|query-tile kernel|keep rate|passed|failed" > $O/gpu_tests_parity_lines.txt
tail -3 $O/gpu_tests_parity_lines.txt; tail -c 300 $O/bench.err
