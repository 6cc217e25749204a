# HBM-side counters of the eager single-stream bench (separate --pmc passes, kernel trace only) + kernel-trace stats of the default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc2/$c -o p -- python $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc2/$c -name "p_results.db" | head -1) --by-grid > $R/gpurun_out/pmc2/${c}.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc2/stats -o s -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/pmc2/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/pmc2/stats -name "s_results.db" | head -1) > $R/gpurun_out/pmc2/kernel_stats.txt 2>&1
head -30 $R/gpurun_out/pmc2/kernel_stats.txt
