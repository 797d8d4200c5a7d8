#!/bin/bash
mkdir -p gpurun_out/lv
timeout 1500 python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_overlap.py tests/test_gpu_mvsnet.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python scripts/dev/lv_probe.py --quick --ab 2>&1 | grep "A/B"
timeout 600 python scripts/dev/lv_probe.py --quick --ab 2>&1 | grep "A/B"
