#!/bin/bash
# round 6, call M: time-chunked perceptual pass beside the tape replay (CADDY_PERC_CHUNKS=1: the one-pass form of rounds 2-5): perceptual parity tests, step A/B over chunk tables
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "perceptual or deterministic or bair256" > gpurun_out/m_model.txt 2>&1; tail -4 gpurun_out/m_model.txt
bash tools/gpu_ab.sh "one pass|CADDY_PERC_CHUNKS=1" "2 chunks, tape2 inline|CADDY_PERC_CHUNKS=2 CADDY_PERC_TAPE2_LATE=0" "2 chunks|CADDY_PERC_CHUNKS=2" "bounds 9|CADDY_PERC_BOUNDS=9" "bounds 6|CADDY_PERC_BOUNDS=6" "bounds 5|CADDY_PERC_BOUNDS=5" "bounds 10,5|CADDY_PERC_BOUNDS=10,5" "bounds 9,3|CADDY_PERC_BOUNDS=9,3" "3 chunks|CADDY_PERC_CHUNKS=3" "one pass|CADDY_PERC_CHUNKS=1" "2 chunks|CADDY_PERC_CHUNKS=2" > /dev/null
cat gpurun_out/ab.txt
