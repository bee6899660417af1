#!/bin/bash
# forks of the backward in batches of n (CADDY_FORK_BATCH): parity at n = 8, then A/B on the E/R/A/D-only step
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
CADDY_FORK_BATCH=8 timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "parity or tight or baseline_geometry_properties or full_geometry" 2>&1 | tail -2
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do for v in 1 4 8 16; do echo "CADDY_FORK_BATCH=$v"; CADDY_FORK_BATCH=$v timeout 300 $B 2>&1 | grep "timed region"; done; done
