#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_gpu.txt 2>&1; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.txt | head -20; grep -A14 "slowest" gpurun_out/pytest_gpu.txt | head -16
