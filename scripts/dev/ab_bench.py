#!/usr/bin/env python3
"""bench.py against another build of libpscv (A/B runs on one GPU box): PSCV_LIB=<path to .so> python scripts/dev/ab_bench.py [bench args]"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from wild_deep_mvs_amd import _lib  # noqa: E402
if os.environ.get("PSCV_LIB"):
    _lib.LIB_PATH = os.environ["PSCV_LIB"]
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
