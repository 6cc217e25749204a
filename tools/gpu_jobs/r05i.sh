#!/bin/bash
# round 5, job i: pe_x3 pipelined schedule variants (stamps + checksums)
O=gpurun_out/r05i; mkdir -p $O
for p in 0 1 2; do MV2D_HIP_LIB=mv2d_amd/lib/variants/libpxpipe$p.so python tools/px_trace.py 2>&1 | grep -v Warn > $O/px_trace_pipe$p.txt; echo pipe $p; cat $O/px_trace_pipe$p.txt | tr '\n' ';'; echo; done
