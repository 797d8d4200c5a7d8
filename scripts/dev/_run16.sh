set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_p; mkdir -p $O
for B in 3 2 4 6; do python scripts/dev/stage_pipeline_probe.py --batch $B 2>&1 | grep "^B =" >> $O/stage_probe.txt; done
python scripts/dev/stage_pipeline_probe.py --batch 3 --tune tail_nbk=4 2>&1 | grep "^B =" | sed 's/^/tail_nbk=4: /' >> $O/stage_probe.txt
python scripts/dev/stage_pipeline_probe.py --batch 3 --tune tail_nbk=8 2>&1 | grep "^B =" | sed 's/^/tail_nbk=8: /' >> $O/stage_probe.txt
cat $O/stage_probe.txt
