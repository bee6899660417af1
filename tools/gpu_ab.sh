#!/bin/bash
# A/B of one environment variable (e.g. CADDY_HIP_LIB=<baseline .so> against the in-tree library, CADDY_STREAMS, CADDY_PRECISION) on the E/R/A/D-only step:  bash tools/gpu_ab.sh VAR a b [tests]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do for v in $2 $3; do echo "$1=$v"; env $1=$v timeout 300 $B 2>&1 | grep "timed region"; done; done
if [ -n "$4" ]; then timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "$4" 2>&1 | grep -E "passed|failed|^E " | head; fi
