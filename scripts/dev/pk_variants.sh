#!/bin/bash
# Diagnostic builds of libpscv.so whose PACKED warp kernel ("warp_tiled" = 3) carries one of the cheap hazard paddings of
# warp_cost_tiled.hip (WL_FIX_NOP / WL_FIX_MOV / WL_FIX_B64): scripts/dev/libpscv_pk_<name>.so (gitignored; travels to the GPU box).
#   bash scripts/dev/pk_variants.sh            # build
#   for v in nop mov b64; do PSCV_LIB=$PWD/scripts/dev/libpscv_pk_$v.so python -m pytest tests/test_gpu_overlap.py -k packed -s -q; done
set -e
cd "$(dirname "$0")/../../wild_deep_mvs_amd/csrc"
OBJS=$(grep "^OBJS" Makefile | sed "s/OBJS *:= *//")
# round 4, second set: which packed instructions does the defect need?  noslp = SLP vectorizer off (only the explicit two-wide final
# expression stays packed); xfinal / xblend / xsums / xblendsums = that part forced scalar, the rest packed
declare -A FLAGS=( [nop]="-DWL_FIX_NOP" [mov]="-DWL_FIX_MOV" [b64]="-DWL_FIX_B64" [noslp]="-fno-slp-vectorize" [xfinal]="-DWL_X_FINAL"
                   [xblend]="-DWL_X_BLEND" [xsums]="-DWL_X_SUMS" [xblendsums]="-DWL_X_BLEND -DWL_X_SUMS" [xall]="-DWL_X_BLEND -DWL_X_SUMS -DWL_X_FINAL"
                   [xallcoords]="-DWL_X_BLEND -DWL_X_SUMS -DWL_X_FINAL -DWL_X_COORDS" [xcoords]="-DWL_X_COORDS" [xbox]="-DWL_X_BOX"
                   [eblend]="-fno-slp-vectorize -DWL_E_BLEND -DWL_X_FINAL" [esums]="-fno-slp-vectorize -DWL_E_SUMS -DWL_X_FINAL"
                   [eblendsums]="-fno-slp-vectorize -DWL_E_BLEND -DWL_E_SUMS -DWL_X_FINAL" [noslpxfinal]="-fno-slp-vectorize -DWL_X_FINAL"
                   [eblendhi]="-fno-slp-vectorize -DWL_E_BLEND -DWL_E_BLEND_HI -DWL_X_FINAL"
                   [xallcoordsbox]="-DWL_X_BLEND -DWL_X_SUMS -DWL_X_FINAL -DWL_X_COORDS -DWL_X_BOX" )
for v in ${VARIANTS:-nop mov b64 noslp xfinal xblend xsums xblendsums xall}; do
    o=/tmp/wl_pk_$v.o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DWL_PK ${FLAGS[$v]} -c warp_cost_tiled.hip -o $o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -o ../../scripts/dev/libpscv_pk_$v.so $(echo "$OBJS" | sed "s#warp_cost_tiled_pk.o#$o#")
    echo built scripts/dev/libpscv_pk_$v.so
done
