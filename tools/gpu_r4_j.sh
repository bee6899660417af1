#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for v in "CADDY_WGRAD_BLOCKS=256" "CADDY_WGRAD_BLOCKS=128" "CADDY_WGRAD_BLOCKS=192" "CADDY_WGRAD_BLOCKS=64" "CADDY_WGRAD_BLOCKS=256" "CADDY_WGRAD_OCC=1" "CADDY_HX_R64=0" "CADDY_HX_R64=1024"; do
  echo "$v:"; env $v timeout 300 $B 2>&1 | grep "timed region"
done
timeout 300 python tools/bench_rollout.py 2>&1 | tail -3
