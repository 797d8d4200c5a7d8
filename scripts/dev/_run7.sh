set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_g; mkdir -p $O
python -m pytest tests/test_gpu_vis.py -q -x -k "homography" > $O/pytest_h.txt 2>&1
tail -8 $O/pytest_h.txt
