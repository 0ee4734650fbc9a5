#!/bin/bash
# Collects the rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> [size] [flow]      <tag> names the output files, e.g. r03; size 512 (default) or 256
# 1. kernel-trace --stats of the bench.py run of that size (80 sweeps)
# 2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass) of the headless C++ bench
# Output under gpurun_out/prof_<tag>_<size>/ ; tools/pmc_to_json.py turns it into profiles/.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r01}
N=${2:-512}
OUT=gpurun_out/prof_${TAG}_$N
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench$N -- python3 bench.py --size $N --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
B=./fluidsolver_amd/bin/fs_bench
# the counter passes run the tile variant, chunking and store policy the bench run above timed as
# best (its JSON line names them), not whatever a timing under the profiler would pick
eval $(python3 - "$OUT/stats.log" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
a = json.loads(line)["autotune"]
deep = (a.get("jacobi_sweeps_per_pass") or 4) > 4
print("export FS_JACOBI_TUNE=0 FS_JACOBI_CFG=%d FS_JACOBI_NT=%d FS_JACOBI_DEEP=%d FS_JACOBI_DEEP_LZ=%d FS_JACOBI_DEEP_NT=%d" % (
    max(a["jacobi_tile_variant"], 0), a["jacobi_nontemporal_stores"], 1 if deep else 0,
    (a.get("jacobi_deep_planes_per_chunk") or 0) if deep else 0, (a.get("jacobi_deep_nontemporal_stores") or 0) if deep else 0))
if a["jacobi_planes_per_chunk"] > 0:
    print("export FS_JACOBI_LZ=%d" % a["jacobi_planes_per_chunk"])
PY
)
echo "counter passes with FS_JACOBI_CFG=$FS_JACOBI_CFG FS_JACOBI_LZ=$FS_JACOBI_LZ FS_JACOBI_NT=$FS_JACOBI_NT FS_JACOBI_DEEP=$FS_JACOBI_DEEP"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $B $N 80 8 2 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $B $N 80 8 2 > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o p -- $B $N 80 8 2 > $OUT/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -o p -- $B $N 80 8 2 > $OUT/pmc_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o p -- $B $N 80 8 2 > $OUT/pmc_l2.log 2>&1
tail -1 $OUT/stats.log | cut -c1-400
find $OUT -name "*.csv" | head -20
