#!/bin/bash
# round-4 GPU call 2: packed-build probe, new / changed tests, bench with alt_geometry
mkdir -p gpurun_out/r4b
O=$PWD/gpurun_out/r4b
timeout 600 python scripts/dev/pk_probe.py > $O/pk_probe.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_warp_cost.py tests/test_gpu_mvsnet.py "tests/test_gpu_fullsize.py::test_mvsnet_fullsize_matches_oracle_on_windows" tests/test_gpu_harness.py tests/test_gpu_vis.py -q -s > $O/tests.txt 2>&1
timeout 900 python bench.py --no-training > $O/bench.json 2> $O/bench.err
tail -n 12 $O/pk_probe.txt; grep -a "passed\|failed\|FAILED" $O/tests.txt | tail -n 20; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4b/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("one_view_at_a_time_ms"))
print(json.dumps(d.get("alt_geometry"), indent=1)[:3000])
print(d["kernels_us"])
PY
