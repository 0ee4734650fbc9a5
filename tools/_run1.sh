set -e
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_slab_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -8
