#!/bin/bash
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4
timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 512 80 5 2
timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 256 80 10 3
timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 128 40 20 3
timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 64 20 100 10
