#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_serial
CADDY_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/bench_serial_erad.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 70 > gpurun_out/mid_step_breakdown_serial_erad.txt; head -3 gpurun_out/mid_step_breakdown_serial_erad.txt
rm -rf gpurun_out/prof_serial
CADDY_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/bench_serial.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 70 > gpurun_out/mid_step_breakdown_serial.txt; head -3 gpurun_out/mid_step_breakdown_serial.txt
rm -rf gpurun_out/prof_serial
