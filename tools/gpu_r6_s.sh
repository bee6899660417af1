#!/bin/bash
# round 6, call S: chunk boundaries of the pipelined perceptual pass (CADDY_PERC_BOUNDS, A/B aid), BAIR
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_ab.sh "10,5 (default)|" "12,6|CADDY_PERC_BOUNDS=12,6" "11,5|CADDY_PERC_BOUNDS=11,5" "12,7,3|CADDY_PERC_BOUNDS=12,7,3" "9,4|CADDY_PERC_BOUNDS=9,4" "10,5 (default)|" "12,6|CADDY_PERC_BOUNDS=12,6" "11,6|CADDY_PERC_BOUNDS=11,6" "8,3|CADDY_PERC_BOUNDS=8,3" > /dev/null
cat gpurun_out/ab.txt
