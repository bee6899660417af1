#!/bin/bash
# round 6, call P: first chunk of the pipelined perceptual pass with its full-resolution level on the main stream (level parallelism kept where nothing runs beside it)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "perceptual or deterministic or bair256" > gpurun_out/p_model.txt 2>&1; tail -3 gpurun_out/p_model.txt
A=$PWD/playablevideogeneration_amd/csrc/build_alt
bash tools/gpu_ab.sh "one pass|CADDY_PERC_CHUNKS=1" "2 chunks|" "3 chunks|CADDY_PERC_CHUNKS=3" "one pass|CADDY_PERC_CHUNKS=1" "2 chunks|" "3 chunks|CADDY_PERC_CHUNKS=3" "4 chunks|CADDY_PERC_CHUNKS=4" > /dev/null
cat gpurun_out/ab.txt
