#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/layer_profile.py > gpurun_out/r3_layers_b.txt 2>&1
grep -E "Cout=    3|total" gpurun_out/r3_layers_b.txt
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2 3; do timeout 300 $B 2>&1 | grep "timed region"; done
