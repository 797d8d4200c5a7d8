set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_r; mkdir -p $O
python scripts/dev/knob_sweep_views.py default warp_ppd=48 warp_ppd=64 sweep_dc=32 sweep_dc=48 sweep_dc=96 tail_nbk=4 tail_nbk=8 tail_nbk=16 s2s_slots=512 s2s_slots=1024 2>&1 | grep -v amdgpu > $O/knobs.txt
cat $O/knobs.txt
