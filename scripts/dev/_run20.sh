set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_t; mkdir -p $O
python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_fullsize.py tests/test_gpu_mvsnet.py -q -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
python scripts/dev/warp_ab.py 4 2>&1 | grep -v amdgpu > $O/warp_ab.txt; cat $O/warp_ab.txt
python bench.py --no-other-configs --no-training --no-live-traffic --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_t/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["repeats"]["ms_per_step_all"], d["kernels_us"]["warp_cost[0]"], d["graph_replay_equals_eager_on_fresh_inputs"], d["alt"]["ms_per_step"])
PY
