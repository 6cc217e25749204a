#!/bin/bash
# round 5, job c: rest of the GPU tests, per-kernel table of the index-exact route, per-phase stamps of pe_x3
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -40 $O/pytest.txt | cut -c1-220
HEAD=34 tools/prof_cmd.sh r05c/prof_exact python tools/run_engine.py --batch 16 --steps 20
for r in 3 5; do MV2D_HIP_LIB=mv2d_amd/lib/variants/libpxtrace$r.so python tools/px_trace.py 2>&1 | grep -v Warn > $O/px_trace_ring$r.txt; cat $O/px_trace_ring$r.txt; done
