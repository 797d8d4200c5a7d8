set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_a; mkdir -p $O
python scripts/dev/step_matrix.py --dtype bf16 --rounds 3 > $O/step_matrix.txt 2>&1
python scripts/dev/wl_residency.py > $O/wl_residency.txt 2>&1
rocprofv3 --kernel-trace -d $O/tl_lock -o t -- python scripts/dev/step_timeline.py run --batch 3 > $O/tl_lock.log 2>&1
python scripts/dev/step_timeline.py show $O/tl_lock --launches-per-step 39 > $O/timeline_lockstep.txt 2>&1
rocprofv3 --kernel-trace -d $O/tl_stag -o t -- python scripts/dev/step_timeline.py run --batch 3 --stagger > $O/tl_stag.log 2>&1
python scripts/dev/step_timeline.py show $O/tl_stag --launches-per-step 39 > $O/timeline_stagger.txt 2>&1
find $O -name "*.db" -size +20M -delete
tail -30 $O/step_matrix.txt; cat $O/wl_residency.txt | tail -8
