#!/bin/bash
# BatchNorm fusion, phase A/B: kernel + model parity on the GPU, then the E/R/A/D-only step with the switches off / on
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_into or hx" 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "parity or tight or full_geometry or baseline_geometry_properties" 2>&1 | tail -3
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
echo "lazy=0 stats=0"; CADDY_BN_LAZY=0 CADDY_BN_EPI_STATS=0 timeout 300 $B 2>&1 | grep "timed region"
echo "lazy=0 stats=1"; CADDY_BN_LAZY=0 CADDY_BN_EPI_STATS=1 timeout 300 $B 2>&1 | grep "timed region"
echo "lazy=1 stats=0"; CADDY_BN_LAZY=1 CADDY_BN_EPI_STATS=0 timeout 300 $B 2>&1 | grep "timed region"
echo "lazy=1 stats=1"; timeout 300 $B 2>&1 | grep "timed region"
done
rm -rf gpurun_out/prof_serial
CADDY_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs > /dev/null 2> gpurun_out/r3c_serial.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 40 > gpurun_out/r3c_erad_breakdown_serial.txt
head -30 gpurun_out/r3c_erad_breakdown_serial.txt
rm -rf gpurun_out/prof_serial
