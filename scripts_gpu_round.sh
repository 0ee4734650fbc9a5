#!/bin/bash
# One GPU session: smoke, headless shim, bench, rocprofv3 kernel stats.  Steps chained; any failure ends it.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log
timeout -k 10 60 ./fluidsolver_amd/bin/fs_headless 5 2>&1 | tee gpurun_out/headless.log
timeout -k 10 300 python -u bench.py --size 256 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tee gpurun_out/bench256.log
timeout -k 10 400 python -u bench.py --size 512 --steps 5 --warmup 2 2>&1 | tee gpurun_out/bench512.log
rm -rf gpurun_out/prof_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o b512 -- python3 bench.py --size 512 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
find gpurun_out/prof_stats -name "*stats*" | head
