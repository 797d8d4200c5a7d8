#!/bin/bash
# round-4 GPU call 6: 5-workgroups-per-CU warp variant A/B, updated tests, bench with the forked-graph default
mkdir -p gpurun_out/r4f
O=$PWD/gpurun_out/r4f
for i in 1 2; do
  echo "== product"; timeout 200 python scripts/wbench.py --only tiled --reps 40
  echo "== wl5 (254-texel arena, launch_bounds(256,5), 96 VGPRs + 36 B scratch)"; PSCV_LIB=$PWD/scripts/dev/libpscv_wl5.so timeout 200 python scripts/wbench.py --only tiled --reps 40
done > $O/wl5.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_mvsnet.py tests/test_gpu_dist.py -q -k "separate_streams or row_slab or stream_mode" > $O/tests.txt 2>&1
timeout 900 python bench.py --no-training --no-other-configs > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --no-training --no-other-configs --no-cpu-baseline --no-live-traffic --batch-mode batched > $O/bench_batched.json 2>> $O/bench.err
PSCV_LIB=$PWD/scripts/dev/libpscv_wl5.so timeout 300 python bench.py --no-training --no-other-configs --no-cpu-baseline --no-live-traffic > $O/bench_wl5.json 2>> $O/bench.err
grep -v amdgpu.ids $O/wl5.txt | cut -c1-200; grep -a "passed\|failed\|FAILED" $O/tests.txt | tail -n 5
python - <<'PY'
import json
for f in ("bench.json","bench_batched.json","bench_wl5.json"):
    try:
        d=json.loads(open("gpurun_out/r4f/"+f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("one_view_at_a_time_ms"), d["kernels_us"].get("warp_cost[0]"), d["timing"][:120])
    except Exception as e:
        print(f, "ERR", e)
PY
