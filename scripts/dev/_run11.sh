set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_k; mkdir -p $O
python scripts/dev/config_launches.py 4 > $O/launches_4.txt 2>&1
python scripts/dev/config_launches.py 3 > $O/launches_3.txt 2>&1
python scripts/dev/config_launches.py 5 > $O/launches_5.txt 2>&1
tail -3 $O/launches_4.txt
