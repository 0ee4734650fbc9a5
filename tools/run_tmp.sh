set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -2
