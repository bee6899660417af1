#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "rccl or native" 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
