set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_i; mkdir -p $O
for B in 2 4; do
rocprofv3 --kernel-trace -d $O/tl_b$B -o t -- python scripts/dev/step_timeline.py run --batch $B > $O/tl_b$B.log 2>&1
python scripts/dev/step_timeline.py show $O/tl_b$B --launches-per-step $((13*B)) > $O/timeline_b$B.txt 2>&1
done
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -d $O/tl_q8 -o t -- python scripts/dev/step_timeline.py run --batch 3 > $O/tl_q8.log 2>&1
python scripts/dev/step_timeline.py show $O/tl_q8 --launches-per-step 39 > $O/timeline_q8.txt 2>&1
find $O -name "*.db" -delete
for f in b2 b4 q8; do echo "== $f"; grep "proj_cams\|softargmin\|last step" $O/timeline_$f.txt | head -12; done
