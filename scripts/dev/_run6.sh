set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_f; mkdir -p $O
python -m pytest tests/test_gpu_mvsnet.py -q -x -k "view_pipeline or stream" > $O/pytest_pipe.txt 2>&1
python bench.py --no-other-configs --no-training --no-live-traffic --no-cpu-baseline > $O/bench_views.json 2> $O/bench_views.err
python bench.py --no-other-configs --no-training --no-live-traffic --no-cpu-baseline --batch-mode streams > $O/bench_streams.json 2> $O/bench_streams.err
tail -5 $O/pytest_pipe.txt
python - <<'PY'
import json
for n in ("views","streams"):
    try:
        d=json.loads(open(f"gpurun_out/r06_f/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["repeats"]["ms_per_step_all"], d["alt"]["ms_per_step"], d["graph_replay_equals_eager_on_fresh_inputs"], d["config"]["batch_mode"])
    except Exception as e:
        print(n, "ERR", e); print(open(f"gpurun_out/r06_f/bench_{n}.err").read()[-2000:])
PY
