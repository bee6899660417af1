#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/layer_profile.py > gpurun_out/r3_layers_b.txt 2>&1
grep -E "phase|total" gpurun_out/r3_layers_b.txt
