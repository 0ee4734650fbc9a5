#!/bin/bash
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 150 --timeout-method thread -k "jacobi or full_step or 64cube" 2>&1 | tail -5
for cfg in 0 1 2 3 4 5; do
  echo "== cfg $cfg"
  FS_JACOBI_CFG=$cfg timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 512 80 3 1
done
for lz in 32 64 128 512; do
  echo "== cfg 0 lz $lz"
  FS_JACOBI_LZ=$lz timeout -k 10 120 ./fluidsolver_amd/bin/fs_bench 512 80 3 1
done
