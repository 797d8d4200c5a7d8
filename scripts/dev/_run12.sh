set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_l; mkdir -p $O
python -m pytest tests/test_gpu_cvp.py -q -x -s -k "forward_parity" > $O/pytest_cvp.txt 2>&1
grep "depth_est_list\|storage-emulated\|passed\|failed\|Error\|assert" $O/pytest_cvp.txt | head -40
