#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "nogumbel" 2>&1 | grep -v "^$" | tail -40 | cut -c1-250
CADDY_BN_LAZY=0 CADDY_BN_EPI_STATS=0 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "nogumbel" 2>&1 | tail -3 | cut -c1-250
