#!/bin/bash
# usage: tools/isa_waits.sh <file.hip> [-Dflags] : per kernel registers / scratch, and every s_waitcnt that (nearly) drains the vector-memory
# queue with what was issued since the previous such wait -- "gload=1 | vmcnt(0)" runs are serialised memory round trips.
src=$1; shift
out=/tmp/isa_$(basename $src .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value "$@" -I$(dirname $0)/../include -S --cuda-device-only -o $out $(dirname $0)/../mv2d_amd/csrc/$src 2>/dev/null
grep -E "^\s+\.(vgpr_count|agpr_count|vgpr_spill_count|private_segment_fixed_size|name):" $out | paste - - - - - | sed 's/\s\+/ /g'
awk '/^[_a-zA-Z].*:$/ && !/^\.L/{k=$1} /v_mfma/{m++} /global_load|buffer_load/{g++} /global_store|buffer_store/{s++} /ds_read/{d++} /s_barrier/{b++}
     /s_waitcnt vmcnt\(0\)|s_waitcnt vmcnt\(1\)$/{ printf "%s line %d: mfma=%d gload=%d gstore=%d dsread=%d barriers=%d | %s\n", k, NR, m,g,s,d,b,$0; m=0;g=0;d=0;s=0;b=0}' $out
