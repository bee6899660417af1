#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "batchnorm or bn or misc or wgrad_hx" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "config_branches or tight" 2>&1 | tail -2
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
echo "occ1"; timeout 300 $B 2>&1 | grep "timed region"
echo "occ2 512"; CADDY_WGRAD_OCC=2 timeout 300 $B 2>&1 | grep "timed region"
echo "occ2 256"; CADDY_WGRAD_OCC=2 CADDY_WGRAD_BLOCKS=256 timeout 300 $B 2>&1 | grep "timed region"
echo "occ2 384"; CADDY_WGRAD_OCC=2 CADDY_WGRAD_BLOCKS=384 timeout 300 $B 2>&1 | grep "timed region"
done
echo serial; CADDY_SIDE_STREAM=0 timeout 300 $B 2>&1 | grep "timed region"
echo serial occ2; CADDY_WGRAD_OCC=2 CADDY_SIDE_STREAM=0 timeout 300 $B 2>&1 | grep "timed region"
