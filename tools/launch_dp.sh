#!/bin/bash
# What the driver's scaling run does on an N-GPU MI355X node (one process per GPU, RCCL over xGMI):   bash tools/launch_dp.sh N [steps] [warmup]
# Prints bench.py's JSON line (rank 0): value = clips/s over all N GPUs (weak scaling: per-GPU batch fixed), rccl_ranks_seen = N.
N=${1:-8}; K=${2:-10}; W=${3:-3}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
SCRIPT=${CADDY_DP_SCRIPT:-bench.py}      # (tests: tests/dp_sim_bench.py runs the same control flow on the host simulator over gloo)
if [ "$N" = "1" ]; then exec python $SCRIPT --gpus 1 --steps $K --warmup $W; fi
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29511} $SCRIPT --gpus $N --steps $K --warmup $W
