mkdir -p gpurun_out/r2
timeout 280 python bench.py --steps 100 > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
timeout 200 python bench.py --steps 100 --workload cfg3_t --no-cpu-baseline > gpurun_out/r2/bench_cfg3_t.json 2>> gpurun_out/r2/bench_default.err
timeout 200 python bench.py --steps 100 --workload cfg5_t --batch 2 --no-cpu-baseline > gpurun_out/r2/bench_cfg5_t.json 2>> gpurun_out/r2/bench_default.err
timeout 200 python bench.py --steps 100 --batch 1 --no-cpu-baseline > gpurun_out/r2/bench_batch1.json 2>> gpurun_out/r2/bench_default.err
timeout 200 python bench.py --steps 100 --batch 1 --inflight 1 --no-cpu-baseline > gpurun_out/r2/bench_single.json 2>> gpurun_out/r2/bench_default.err
tail -c 600 gpurun_out/r2/bench_default.err
