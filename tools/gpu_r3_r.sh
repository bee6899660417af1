#!/bin/bash
# 256-channel-block 8-wave conv_hx tile (CADDY_HX_WIDE): perceptual-loss parity, then A/B of the full step
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
CADDY_HX_WIDE=1 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "perceptual" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-rollout --no-extra-legs --no-plugin --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do for v in 0 1; do echo "CADDY_HX_WIDE=$v"; CADDY_HX_WIDE=$v timeout 300 $B 2>&1 | grep "timed region"; done; done
