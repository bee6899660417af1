#!/bin/bash
# sweep one environment switch on the E/R/A/D-only step:  bash tools/gpu_sweep.sh VAR v1 v2 ...
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
VAR=$1; shift
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do for v in "$@"; do echo "$VAR=$v"; env $VAR=$v timeout 300 $B 2>&1 | grep "timed region"; done; done
