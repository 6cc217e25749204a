#!/bin/bash
# round 6: wave = head formulation of the shared-tile cross attention; tr16 probe; timing split (no arithmetic / no DMA variants)
O=gpurun_out/r06b; mkdir -p $O
hipcc --offload-arch=gfx950 tools/probes/tr16_probe.hip -o /tmp/tr16_probe 2>/dev/null && /tmp/tr16_probe | tail -9
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "xattn_group" 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_engine.py -q -x 2>&1 | tail -5
for w in "cfg3_t --batch 16" "cfg5_t --batch 4"; do
  set -- $w; n=$1
  timeout 300 python tools/run_engine.py --workload "$@" --steps 20 2>&1 | tail -1
  for v in xg_nocompute xg_nodma; do
    [ -f mv2d_amd/lib/variants/lib$v.so ] && MV2D_HIP_LIB=mv2d_amd/lib/variants/lib$v.so timeout 300 python tools/run_engine.py --workload "$@" --steps 20 2>&1 | tail -1 | sed "s/^/[$v] /"
  done
  HEAD=6 tools/prof_cmd.sh r06b/prof_$n python tools/run_engine.py --workload "$@" --steps 10 2>&1 | tail -5
done
