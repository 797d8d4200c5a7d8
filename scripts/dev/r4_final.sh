#!/bin/bash
# round-4 final GPU pass: whole GPU suite, driver-style bench line, rocprofv3 kernel stats + counter passes of the final build
mkdir -p gpurun_out/r4z
O=$PWD/gpurun_out/r4z
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
bash scripts/profile.sh r04 > $O/profile.log 2>&1
cp gpurun_out/prof_r04/summary.txt $O/prof_summary.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04/stats -o bench -- python bench.py --batch 1 --eager --no-other-configs --no-training --no-cpu-baseline --no-live-traffic > $O/stats.log 2>&1
find gpurun_out/prof_r04/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find gpurun_out/prof_r04 -name "*.db" -size +5M -delete
tail -n 4 $O/gpu_tests.txt | cut -c1-300; cat $O/smoke.txt | tail -n 2; head -c 400 $O/bench.json; echo; head -n 30 $O/prof_summary.txt | cut -c1-200
