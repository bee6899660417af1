#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_tl
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tl -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs > /dev/null 2> gpurun_out/r3f.err
python tools/timeline.py gpurun_out/prof_tl/bair_results.db 60 > gpurun_out/r3f_timeline.txt 2>&1
head -80 gpurun_out/r3f_timeline.txt
rm -rf gpurun_out/prof_tl
