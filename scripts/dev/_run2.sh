set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_b; mkdir -p $O
python scripts/dev/view_graphs.py --batch 3 > $O/view_graphs_b3.txt 2>&1
python scripts/dev/view_graphs.py --batch 4 --rounds 2 > $O/view_graphs_b4.txt 2>&1
python scripts/dev/view_graphs.py --batch 2 --rounds 2 > $O/view_graphs_b2.txt 2>&1
rocprofv3 --kernel-trace -d $O/tl_views -o t -- python scripts/dev/view_graphs.py --batch 3 --trace views > $O/tl_views.log 2>&1
python scripts/dev/step_timeline.py show $O/tl_views --launches-per-step 78 > $O/timeline_views.txt 2>&1
find $O -name "*.db" -size +20M -delete
cat $O/view_graphs_b3.txt $O/view_graphs_b4.txt $O/view_graphs_b2.txt
