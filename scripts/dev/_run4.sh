set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_d; mkdir -p $O
python scripts/dev/pair_corun.py > $O/pair_corun.txt 2>&1
python bench.py --no-other-configs --no-training --no-live-traffic > $O/bench_quick.json 2> $O/bench_quick.err
grep -v amdgpu.ids $O/pair_corun.txt; tail -c 3000 $O/bench_quick.json; tail -5 $O/bench_quick.err
