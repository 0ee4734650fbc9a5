#!/bin/bash
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 800 python -u -m pytest tests/test_slab_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -40
