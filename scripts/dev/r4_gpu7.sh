#!/bin/bash
# round-4 GPU call 7: warp variants without spills (rf^2 recomputed per trip): 4 and 5 workgroups per CU
mkdir -p gpurun_out/r4g
O=$PWD/gpurun_out/r4g
for i in 1 2; do
  for v in product wl4b wl5b; do
    echo "== $v"
    if [ $v = product ]; then timeout 200 python scripts/wbench.py --only tiled --reps 40; else PSCV_LIB=$PWD/scripts/dev/libpscv_$v.so timeout 200 python scripts/wbench.py --only tiled --reps 40; fi
  done
done > $O/wl.txt 2>&1
for v in product wl4b wl5b; do
  if [ $v = product ]; then L=""; else L=$PWD/scripts/dev/libpscv_$v.so; fi
  PSCV_LIB=$L timeout 300 python bench.py --no-training --no-other-configs --no-cpu-baseline --no-live-traffic > $O/bench_$v.json 2>> $O/bench.err
done
grep -v amdgpu.ids $O/wl.txt | cut -c1-200
python - <<'PY'
import json
for v in ("product","wl4b","wl5b"):
    try:
        d=json.loads(open(f"gpurun_out/r4g/bench_{v}.json").read().strip().splitlines()[-1])
        print(v, d["value"], d["ms_per_step"], d["config"].get("one_view_at_a_time_ms"), d["kernels_us"].get("warp_cost[0]"))
    except Exception as e:
        print(v, "ERR", e)
PY
