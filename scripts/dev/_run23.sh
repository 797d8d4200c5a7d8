set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_w; mkdir -p $O
python scripts/dev/prio_probe.py 2>&1 | grep -v amdgpu > $O/prio.txt; cat $O/prio.txt
