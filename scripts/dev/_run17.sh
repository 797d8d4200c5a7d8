set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_q; mkdir -p $O
bash scripts/profile.sh r06 > $O/profile.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
tail -c 300 $O/bench_full.json; ls gpurun_out/prof_r06 gpurun_out/prof_r06/trace | head -30
