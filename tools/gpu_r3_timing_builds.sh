#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
echo base; timeout 300 $B 2>&1 | grep "timed region"
for e in 1 2 4 5; do echo "EXP=$e (1 no weight-tile traffic, 2 no activation staging, 4 no per-tap barrier)"; CADDY_HIP_LIB=$PWD/playablevideogeneration_amd/csrc/build/exp/libcaddy_exp$e.so timeout 300 $B 2>&1 | grep "timed region"; done
echo base; timeout 300 $B 2>&1 | grep "timed region"
