#!/bin/bash
# round-4 GPU call 5: row-slab shard with the exact row offset, full suite, streams / graph probe
mkdir -p gpurun_out/r4e
O=$PWD/gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_warp_cost.py -q -s -k "row_slab or bench_sharded or rccl_backend" > $O/dist.txt 2>&1
timeout 600 python scripts/dev/streams_graph_probe.py > $O/streams.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
grep -a "parity\] row\|bench sharded\|passed\|failed\|FAILED\|Error" $O/dist.txt | cut -c1-500 | tail -n 24
cat $O/streams.txt | grep -v amdgpu.ids | cut -c1-400
tail -n 6 $O/gpu_tests.txt | cut -c1-300
