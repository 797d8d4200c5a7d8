set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_h; mkdir -p $O
python scripts/dev/small_layers.py > $O/small_nt.txt 2>&1
python scripts/dev/small_layers.py --batch 3 > $O/small_nt_b3.txt 2>&1
grep -v amdgpu $O/small_nt.txt; grep -v amdgpu $O/small_nt_b3.txt
