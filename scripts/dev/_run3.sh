set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c; mkdir -p $O
for pad in 0 5 14 42; do
  echo "== warp_lds_pad=$pad" >> $O/views_pad.txt
  python scripts/dev/view_graphs.py --batch 3 --rounds 2 --tune warp_lds_pad=$pad >> $O/views_pad.txt 2>&1
done
echo "== B=6 pad 0" >> $O/views_pad.txt
python scripts/dev/view_graphs.py --batch 6 --rounds 2 >> $O/views_pad.txt 2>&1
echo "== B=6 pad 5" >> $O/views_pad.txt
python scripts/dev/view_graphs.py --batch 6 --rounds 2 --tune warp_lds_pad=5 >> $O/views_pad.txt 2>&1
grep -v amdgpu.ids $O/views_pad.txt
