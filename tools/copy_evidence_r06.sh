#!/bin/bash
# Copies the files of one tools/r06_evidence.sh run (gpurun_out/r06) into profiles/r06_* and rebuilds profiles/pmc_traffic.json for the four workloads.
# usage: bash tools/copy_evidence.sh "<provenance note>"
cd "$(dirname "$0")/.."
O=gpurun_out/r06
for f in soak_bench_cfg2s.json ablate_exact.txt bench_cfg2s_one_rank_rccl.json default_bench_cfg2s.json default_bench_cfg2s_kernel_stats.txt default_bench_cfg2s_under_rocprof.json \
         driver_shape_bench_cfg2s.json engine_exact_cfg2s_batch1_kernel_stats.txt engine_exact_cfg2s_kernel_stats.txt engine_key16_cfg2s_kernel_stats.txt \
         engine_exact_cfg3t_kernel_stats.txt engine_exact_cfg5t_kernel_stats.txt engine_exact_cfg2s_nc6_kernel_stats.txt engine_optin_group_xattn_cfg3t_kernel_stats.txt \
         engine_optin_pe_rows_in_waves_cfg3t_kernel_stats.txt gpu_tests_parity_lines.txt cluster_order_and_shared_tiles.txt pe_kernel_shapes.txt \
         xattn_fused_queries_per_block.txt pmc_xattn_group_cfg3t.txt train_step_cfg2s.json train_step_cfg3t.json; do cp $O/$f profiles/r06_$f; done
cp $O/prof_rccl/kernel_stats.txt profiles/r06_bench_cfg2s_one_rank_rccl_kernel_stats.txt
for w in cfg2s cfg2s_nc6 cfg3t cfg5t cfg2s_key16 cfg2s_nchw; do for c in $O/pmc_$w/*.txt; do cp "$c" "profiles/r06_pmc_${w}_$(basename "$c")"; done; done
N="${1:-rocprofv3 --kernel-trace --pmc (separate passes per counter), tools/pmc_bench.sh via tools/r06_evidence.sh, MI355X, round 6, index-exact route}"
python tools/pmc_to_json.py $O/pmc_cfg2s cfg2_s@16 "$N" > /dev/null
python tools/pmc_to_json.py $O/pmc_cfg2s_nc6 cfg2_s_nc6@16 "$N" > /dev/null
python tools/pmc_to_json.py $O/pmc_cfg3t cfg3_t@16 "$N" > /dev/null
python tools/pmc_to_json.py $O/pmc_cfg5t cfg5_t@4 "$N" > /dev/null
python tools/pmc_to_json.py $O/pmc_cfg2s_key16 cfg2_s@16:key16 "$N (opt-in key16 mode)" > /dev/null
for w in "cfg2s 16" "cfg2s_nc6 16" "cfg3t 16" "cfg5t 4" "cfg2s_key16 16" "cfg2s_nchw 16"; do set -- $w; python tools/pmc_whole_path.py $O/pmc_$1 $2; done > profiles/r06_whole_path_traffic.txt
grep "MB per sample" profiles/r06_whole_path_traffic.txt
