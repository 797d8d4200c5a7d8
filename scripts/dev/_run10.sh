set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_j; mkdir -p $O
for ppd in 32 24 16 40 48; do
  echo "== warp_ppd=$ppd" >> $O/ppd.txt
  python scripts/dev/view_graphs.py --batch 3 --rounds 2 --tune warp_ppd=$ppd 2>&1 | grep "views \|forked " >> $O/ppd.txt
done
cat $O/ppd.txt
