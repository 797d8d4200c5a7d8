#!/bin/bash
timeout 600 python scripts/dev/gc_modes.py 5 2>&1 | grep -v amdgpu.ids | tail -12
