set -e
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 1000 python -u -m pytest tests -x -v -m gpu --timeout 250 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { echo "exit $?"; tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json
cut -c1-400 gpurun_out/bench_default.json
