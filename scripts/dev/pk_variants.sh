#!/bin/bash
# Diagnostic builds of libpscv.so whose PACKED warp kernel ("warp_tiled" = 3) carries one of the cheap hazard paddings of
# warp_cost_tiled.hip (WL_FIX_NOP / WL_FIX_MOV / WL_FIX_B64): scripts/dev/libpscv_pk_<name>.so (gitignored; travels to the GPU box).
#   bash scripts/dev/pk_variants.sh            # build
#   for v in nop mov b64; do PSCV_LIB=$PWD/scripts/dev/libpscv_pk_$v.so python -m pytest tests/test_gpu_overlap.py -k packed -s -q; done
set -e
cd "$(dirname "$0")/../../wild_deep_mvs_amd/csrc"
OBJS=$(grep "^OBJS" Makefile | sed "s/OBJS *:= *//")
for v in nop mov b64; do
    V=$(echo $v | tr a-z A-Z)
    o=/tmp/wl_pk_$v.o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DWL_PK -DWL_FIX_$V -c warp_cost_tiled.hip -o $o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -o ../../scripts/dev/libpscv_pk_$v.so $(echo "$OBJS" | sed "s#warp_cost_tiled_pk.o#$o#")
    echo built scripts/dev/libpscv_pk_$v.so
done
