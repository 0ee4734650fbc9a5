#!/bin/bash
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 150 --timeout-method thread -k "advect or stage or full_step or quirk" 2>&1 | tail -3
for g in 0 2 1; do
echo "== global mask $g"
FS_ADVECT_GLOBAL=$g timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 512 8 3 1
done
