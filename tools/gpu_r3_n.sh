#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
echo "--- prefetcher"; timeout 600 python tools/plugin_profile.py 2>&1 | grep step | tail -3
echo "--- no prefetcher (resident batch)"; PF=0 timeout 600 python tools/plugin_profile.py 2>&1 | grep step | tail -2
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do timeout 300 $B 2>&1 | grep "timed region"; done
timeout 900 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep -E "timed|plugin|roll-out"
