set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 280 --timeout-method thread 2>&1 | tail -4
for a in "512 80 5 2" "256 80 10 3"; do timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench $a | python3 -c "import sys,json; d=json.load(sys.stdin); s=d['stage_ms_per_step']; print(d['n'], d['ms_per_step'], s['advect_velocity'], s['advect_scalars'], s['jacobi'])"; done
