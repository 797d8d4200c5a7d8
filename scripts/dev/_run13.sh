set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_m; mkdir -p $O
( time python -m pytest tests -q -x -m gpu ) > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
tail -c 600 $O/bench_full.json; tail -3 $O/bench_full.err
