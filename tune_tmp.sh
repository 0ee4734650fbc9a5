#!/bin/bash
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== N=1 plain"
timeout -k 10 300 python -u bench.py --size 256 --steps 5 --warmup 2 --no-cpu-baseline
echo "== N=2 on one GPU (RCCL expected to refuse a duplicate GPU -> host fallback)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29700 bench.py --gpus 2 --size 128 --steps 3 --warmup 1 2>&1 | tail -25
