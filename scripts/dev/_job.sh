set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_final2; mkdir -p $O
( time python -m pytest tests -q -x -m gpu ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
tail -c 150 $O/bench_full.json; tail -2 $O/bench_full.err
