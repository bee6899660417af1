#!/bin/bash
# round 6, call O: repeat the checkpoint-resume bit-identity test (a last-bit difference of the fp64 loss sum was seen once), the host-API GPU tests
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_host_api_gpu.py -m gpu -q -x -k "checkpoint_loaded_before_cuda" 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_host_api_gpu.py -m gpu -q 2>&1 | tail -2
