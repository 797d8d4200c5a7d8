#!/bin/bash
mkdir -p gpurun_out/lv
echo "== default"; timeout 600 python scripts/dev/lv_probe.py --quick --ab 2>&1 | grep "A/B\|differing"
echo "== v_fma_f32 (VOP3) build"; PSCV_LIB=$PWD/scripts/dev/libpscv_v3.so timeout 600 python scripts/dev/lv_probe.py --quick --ab 2>&1 | grep "A/B\|differing"
echo "== default"; timeout 600 python scripts/dev/lv_probe.py --quick --ab 2>&1 | grep "A/B\|differing"
echo "== v_fma_f32 (VOP3) build"; PSCV_LIB=$PWD/scripts/dev/libpscv_v3.so timeout 600 python scripts/dev/lv_probe.py --quick --ab 2>&1 | grep "A/B\|differing"
