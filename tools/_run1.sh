set -e
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "mgpcg or headless" --timeout 300 --timeout-method thread 2>&1 | tail -5
for n in 64 128 256 512; do
  timeout -k 10 120 fluidsolver_amd/bin/fs_bench $n 80 6 2 - 0 1 | tail -1 | cut -c1-100
done
timeout -k 10 60 fluidsolver_amd/bin/fs_headless 20 shipped | tail -3
