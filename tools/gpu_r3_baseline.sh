#!/bin/bash
# round-3 baseline: serialised kernel trace of the E/R/A/D-only step (no VGG19 term), per kernel and per (kernel, grid); per-layer conv timing
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_serial
CADDY_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual > gpurun_out/r3_serial.json 2> gpurun_out/r3_serial.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 80 > gpurun_out/r3_erad_breakdown_serial.txt
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 150 --grid > gpurun_out/r3_erad_breakdown_serial_grid.txt
head -5 gpurun_out/r3_erad_breakdown_serial.txt
rm -rf gpurun_out/prof_serial
timeout 300 python tools/layer_profile.py > gpurun_out/r3_layers.txt 2>&1; head -3 gpurun_out/r3_layers.txt
timeout 300 python bench.py --no-perceptual --no-cpu-baseline --no-rollout > gpurun_out/r3_noperc0.json 2> gpurun_out/r3_noperc0.err; grep "timed region" gpurun_out/r3_noperc0.err
