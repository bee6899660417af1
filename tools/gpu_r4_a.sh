#!/bin/bash
# Round 4, call A: grid-barrier probe, point-wise kernel A/B (r3 library vs this tree), E/R/A/D-only step A/B, kernel parity subset -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/_ab/grid_barrier > gpurun_out/r04_grid_barrier.txt 2>&1; cat gpurun_out/r04_grid_barrier.txt | cut -c1-260
timeout 300 python tools/bench_pointwise.py tools/_ab/libcaddy_hip_r3.so playablevideogeneration_amd/csrc/libcaddy_hip.so > gpurun_out/r04_pointwise_ab.txt 2>&1; cat gpurun_out/r04_pointwise_ab.txt | cut -c1-200
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
  echo "r3 lib:"; CADDY_HIP_LIB=$PWD/tools/_ab/libcaddy_hip_r3.so timeout 300 $B 2>&1 | grep "timed region"
  echo "this tree:"; timeout 300 $B 2>&1 | grep "timed region"
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
