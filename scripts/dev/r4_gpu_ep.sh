#!/bin/bash
# explicit-packed (low-half broadcasts only, SLP off) product candidate of the LDS warp kernel vs the scalar product: soak + A/B
mkdir -p gpurun_out/r4k
O=$PWD/gpurun_out/r4k
EP=$PWD/scripts/dev/libpscv_ep.so
PSCV_LIB=$EP timeout 600 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_warp_cost.py -q -s -k "warp or lds or tiled or staged or slab" > $O/ep_tests.txt 2>&1
for i in 1 2; do
  for v in product ep; do
    if [ $v = product ]; then L=""; else L=$EP; fi
    echo "== $v"; PSCV_LIB=$L timeout 200 python scripts/wbench.py --only tiled --reps 40
    PSCV_LIB=$L timeout 300 python bench.py --no-training --no-other-configs --no-cpu-baseline --no-live-traffic > $O/bench_${v}_$i.json 2>> $O/bench.err
  done
done > $O/wbench.txt 2>&1
grep -a "overlap\]\|passed\|failed" $O/ep_tests.txt | cut -c1-200 | tail -n 12; grep -v amdgpu $O/wbench.txt | cut -c1-160
python - <<'PY'
import json
for v in ("product_1","ep_1","product_2","ep_2"):
    d=json.loads(open(f"gpurun_out/r4k/bench_{v}.json").read().strip().splitlines()[-1])
    print(v, d["value"], d["ms_per_step"], d["config"].get("one_view_at_a_time_ms"), d["kernels_us"].get("warp_cost[0]"))
PY
