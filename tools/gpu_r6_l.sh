#!/bin/bash
# round 6, call L: where in the chunk does k_conv_hx<BG> request the next halo tile?  (behind tap 3 = default build, tap 1, tap 0: alternative builds, tools/build_exp.sh)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
A=$PWD/playablevideogeneration_amd/csrc/build_alt
BENCH_ONLY="R " BENCH_REPS=50 timeout 900 python tools/bench_step_convs.py playablevideogeneration_amd/csrc/libcaddy_hip.so $A/libtap1.so $A/libtap0.so 2>&1 | grep -v amdgpu.ids | grep -v wgrad > gpurun_out/l_shapes.txt
cat gpurun_out/l_shapes.txt
bash tools/gpu_ab.sh "tap 3|" "tap 0|CADDY_HIP_LIB=$A/libtap0.so" "tap 1|CADDY_HIP_LIB=$A/libtap1.so" "tap 3|" "tap 0|CADDY_HIP_LIB=$A/libtap0.so" "tap 1|CADDY_HIP_LIB=$A/libtap1.so" > /dev/null
cat gpurun_out/ab.txt
