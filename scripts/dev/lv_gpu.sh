#!/bin/bash
mkdir -p gpurun_out/lv
timeout 600 python scripts/dev/lv_probe.py --quick --stagger 2>&1 | grep "stagger"
