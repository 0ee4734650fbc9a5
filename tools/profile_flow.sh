#!/bin/bash
# Counter passes of bench.py on the plume (BASELINE's workload) and on the developed swirl, to see
# what the gather kernels pay for on a developed flow (run through gpurun):
#   tools/profile_flow.sh <tag> [size]
# Output under gpurun_out/flow_<tag>/ ; tools/pmc_flow_summary.py turns it into profiles/<tag>_flow_counters.json.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r03}
N=${2:-512}
OUT=gpurun_out/flow_$TAG
rm -rf $OUT && mkdir -p $OUT
for FLOW in plume swirl; do
  timeout -s KILL 170 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/${FLOW}_a -o p -- python3 bench.py --size $N --steps 3 --warmup 2 --no-cpu-baseline --flow $FLOW > $OUT/${FLOW}_a.log 2>&1
  timeout -s KILL 170 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${FLOW}_b -o p -- python3 bench.py --size $N --steps 3 --warmup 2 --no-cpu-baseline --flow $FLOW > $OUT/${FLOW}_b.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
