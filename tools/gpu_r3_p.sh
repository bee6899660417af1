#!/bin/bash
# merged weight (un)packing launch: parity subset, A/B on the E/R/A/D-only and full step, then a fresh serial breakdown + stream timeline of the E/R/A/D-only step
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "parity or tight or baseline_geometry_properties or split_operand" 2>&1 | tail -3
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do for v in 0 1; do echo "CADDY_PACK_MERGED=$v"; CADDY_PACK_MERGED=$v timeout 300 $B 2>&1 | grep "timed region"; done; done
rm -rf gpurun_out/prof_serial
CADDY_SIDE_STREAM=0 CADDY_D_STREAM=0 CADDY_AUX_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs > /dev/null 2> gpurun_out/r3p_serial.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 60 > gpurun_out/r3p_erad_breakdown_serial.txt
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 80 --grid > gpurun_out/r3p_erad_breakdown_serial_grid.txt
head -4 gpurun_out/r3p_erad_breakdown_serial.txt
rm -rf gpurun_out/prof_serial gpurun_out/prof_tl
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tl -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs > /dev/null 2> gpurun_out/r3p_tl.err
python tools/timeline.py gpurun_out/prof_tl/bair_results.db 60 > gpurun_out/r3p_timeline.txt 2>&1
head -30 gpurun_out/r3p_timeline.txt
rm -rf gpurun_out/prof_tl
