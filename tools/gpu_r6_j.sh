#!/bin/bash
# round 6, call J: where does a (tap, chunk) step of the register-weights tile variants go?  Timing-study builds (tools/build_exp.sh: HX_EXP bits 1 no halo conversion / store,
# 2 no weight loads, 4 no fragment reads, 8 no halo loads) on the isolated R shapes
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
A=playablevideogeneration_amd/csrc/build_alt
BENCH_ONLY="R " BENCH_REPS=50 timeout 900 python tools/bench_step_convs.py playablevideogeneration_amd/csrc/libcaddy_hip.so $A/libexp1.so $A/libexp8.so $A/libexp9.so 2>&1 | grep -v amdgpu.ids | grep -v wgrad > gpurun_out/j_shapes.txt
cut -c1-40,41-400 gpurun_out/j_shapes.txt
