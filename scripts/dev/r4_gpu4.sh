#!/bin/bash
# round-4 GPU call 4: whole GPU suite (incl. the row-slab shard), bench line
mkdir -p gpurun_out/r4d
O=$PWD/gpurun_out/r4d
timeout 600 python -m pytest tests/test_gpu_dist.py -q -x -s -k "row_slab or bench_sharded or rccl_backend" > $O/dist.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
grep -a "parity\] row\|bench sharded\|passed\|failed\|FAILED\|Error" $O/dist.txt | cut -c1-400 | tail -n 20
tail -n 5 $O/gpu_tests.txt | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4d/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("one_view_at_a_time_ms"), d["roofline"]["frac"], d["roofline"]["avg_us"])
print([ (c["config"], round(c["ms_per_forward"],3)) for c in d["other_configs"]], d["other_configs"][3].get("coarse_48_planes"))
print(d["training"])
PY
