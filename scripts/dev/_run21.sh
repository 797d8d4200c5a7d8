set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_u; mkdir -p $O
python -m pytest tests/test_gpu_warp_cost.py tests/test_gpu_fullsize.py tests/test_gpu_mvsnet.py tests/test_gpu_overlap.py tests/test_gpu_dist.py -q -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
