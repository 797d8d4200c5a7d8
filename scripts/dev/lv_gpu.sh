#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_vis.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -4
