#!/bin/bash
# round 6, call T: the GPU suite three times (flakiness hunt: the driver runs it once, with -x)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_rep$i.txt 2>&1; tail -2 gpurun_out/pytest_gpu_rep$i.txt; done
