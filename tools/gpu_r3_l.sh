#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_host_api_gpu.py -m gpu -x -q -k "split_operand or with_perceptual or rollout or trainer or plugin" 2>&1 | grep -E "passed|failed|^E " | head -10
timeout 300 python tools/bench_rollout.py 2>&1 | grep run
timeout 900 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep -E "timed|erad|plugin|roll-out|exact"
cat gpurun_out/split_vs_exact.json | head -60
