#!/bin/bash
for p in 32 48 64; do echo "== ppd $p"; timeout 600 python scripts/dev/gc_feasibility.py --ppd=$p 2>&1 | grep "groupcorr HOMOG, LDS" | cut -c1-220; done
