#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
echo "--- plugin without a roll-out before it"; timeout 900 python bench.py --no-cpu-baseline --no-rollout 2>&1 >/dev/null | grep -E "timed|plugin"
echo "--- default order (roll-out, then plugin)"; timeout 900 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep -E "timed|plugin|roll-out"
echo "--- GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 900 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep -E "timed|plugin|roll-out"
echo "--- GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 900 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep -E "timed|plugin|roll-out"
