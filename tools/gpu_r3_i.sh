#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
for v in 0 1 2; do echo "BDIRECT=$v"; CADDY_HX_BDIRECT=$v timeout 300 $B 2>&1 | grep "timed region"; done
done
