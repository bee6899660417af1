#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do echo "pool D=1"; CADDY_HX_POOL_D1=1 timeout 300 $B 2>&1 | grep "timed region"; echo "pool D=3";  timeout 300 $B 2>&1 | grep "timed region"; done
