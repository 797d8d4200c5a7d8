set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_e; mkdir -p $O
python scripts/dev/batched_stages.py > $O/batched_stages.txt 2>&1
python scripts/dev/pair_corun.py --tune warp_lds_pad=14 > $O/pair_corun_pad14.txt 2>&1
grep -v amdgpu.ids $O/batched_stages.txt; grep -v amdgpu.ids $O/pair_corun_pad14.txt | head -8
