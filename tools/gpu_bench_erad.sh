#!/bin/bash
# E/R/A/D-only step time (two repetitions) + optional parity subset:  bash tools/gpu_bench_erad.sh [tests]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do timeout 300 $B 2>&1 | grep "timed region"; done
if [ -n "$1" ]; then timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "$1" 2>&1 | tail -2; fi
