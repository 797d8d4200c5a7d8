#!/bin/bash
# round-4 GPU call 3: sub-chunked LDS warp kernel -- parity, overlap, bench with both rigs
mkdir -p gpurun_out/r4c
O=$PWD/gpurun_out/r4c
timeout 1500 python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_mvsnet.py "tests/test_gpu_fullsize.py::test_mvsnet_fullsize_matches_oracle_on_windows" tests/test_gpu_overlap.py -k "not conv2d and not conv3d and not block8 and not backward and not tail" -q > $O/tests.txt 2>&1
timeout 900 python bench.py --no-training > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --no-training --no-other-configs --no-cpu-baseline --no-live-traffic --tune warp_tile=2 > $O/bench_nosplit.json 2>> $O/bench.err
grep -a "passed\|failed\|FAILED" $O/tests.txt | tail -n 20; python - <<'PY'
import json
for f in ("bench.json", "bench_nosplit.json"):
    d=json.loads(open("gpurun_out/r4c/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"].get("one_view_at_a_time_ms"), d["kernels_us"].get("warp_cost[0]"))
    g=d.get("alt_geometry")
    if g:
        for rig in ("probe","dtu"):
            print(rig, g[rig]["warp_cost_us"], g[rig]["hot_path_eager_ms_per_view"], g[rig]["staging_modes_share_of_block_views"])
PY
