#!/bin/bash
# A/B builds of libpscv under another name: scripts/dev/libpscv_<name>.so (gitignored; travels to the GPU box), rebuilding the
# listed sources with extra flags and linking them with the regular objects of the others.
#   bash scripts/dev/ab_build.sh <name> "<file1.hip file2.hip ...>" [extra hipcc flags]
# Compare with: PSCV_LIB=$PWD/scripts/dev/libpscv_<name>.so python scripts/wbench.py / kbench.py / dev/phase_prof.py
set -e
cd "$(dirname "$0")/../../wild_deep_mvs_amd/csrc"
name=$1; files=$2; shift; shift
OBJS=$(grep "^OBJS" Makefile | sed "s/OBJS *:= *//")
for f in $files; do
    mkdir -p ../../gpurun_out/ab
    o=../../gpurun_out/ab/ab_${name}_${f%.hip}.o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w "$@" -c $f -o $o
    OBJS=$(echo "$OBJS" | sed "s#\b${f%.hip}.o#$o#")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -o ../../scripts/dev/libpscv_$name.so $OBJS
echo built scripts/dev/libpscv_$name.so
