set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_n; mkdir -p $O
python scripts/dev/warp_ab.py 5 > $O/warp_ab.txt 2>&1
python -m pytest tests/test_gpu_warp_cost.py -q -x -s -k "adaptive_split" > $O/pytest_split.txt 2>&1
python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_fullsize.py -q -x > $O/pytest_warp.txt 2>&1
grep -v amdgpu $O/warp_ab.txt; grep "parity\] DTU\|passed\|failed\|Error" $O/pytest_split.txt; tail -4 $O/pytest_warp.txt
