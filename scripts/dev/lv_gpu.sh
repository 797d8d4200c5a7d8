#!/bin/bash
mkdir -p gpurun_out/lv
timeout 900 python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_overlap.py -m gpu -q -x -k "row_slab or groupcorr" 2>&1 | tail -3
B="python bench.py --no-other-configs --no-training --no-cpu-baseline --no-live-traffic --steps 60 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step ms', round(d['ms_per_step'], 4), 'one view ms', round(d['config'].get('one_view_at_a_time_ms', 0), 4), 'warp us', round(d['roofline']['avg_us'], 1), 'conv0', d['kernels_us'].get('conv3d[32->8,k3]'))"
done
