#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
for v in 0 1; do echo "AUX_STREAM=$v"; CADDY_AUX_STREAM=$v timeout 300 $B 2>&1 | grep "timed region"; done
done
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "parity or tight or baseline_geometry_properties" 2>&1 | tail -2
