#!/bin/bash
# split-K slab reduce inside the point-wise consumer (roll-out): A/B of the roll-out rate per consumer kind (bit 0 ConvLSTM cell, 1 pool, 2 bilinear x2)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do for v in 0 1 2 4 3; do echo "CADDY_SPLIT_DEFER=$v"; CADDY_SPLIT_DEFER=$v timeout 300 python tools/bench_rollout.py 36 2>&1 | grep "run 2"; done; done
