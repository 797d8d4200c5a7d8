#!/bin/bash
# A/B builds of libpscv with the CURRENT csrc/warp_cost_tiled.hip under another name: scripts/dev/libpscv_<name>.so
# (gitignored; travels to the GPU box).  Compare with: PSCV_LIB=$PWD/scripts/dev/libpscv_<name>.so python scripts/wbench.py
set -e
cd "$(dirname "$0")/../../wild_deep_mvs_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c warp_cost_tiled.hip -o /tmp/wt_$name.o
OBJS=$(grep "^OBJS" Makefile | sed "s/OBJS *:= *//; s#warp_cost_tiled.o#/tmp/wt_$name.o#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -o ../../scripts/dev/libpscv_$name.so $OBJS
echo built scripts/dev/libpscv_$name.so
