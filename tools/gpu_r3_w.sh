#!/bin/bash
# deferred loss read-back in train_epoch: host-API tests, then the plugin leg of the bench (three repetitions)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_api_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-rollout --steps 8 --warmup 2 --profile-steps 0 2>&1 | grep -E "timed region|plugin"; done
