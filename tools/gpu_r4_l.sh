#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_ONLY=final timeout 300 python tools/bench_step_convs.py 2>&1 | tail -4
BENCH_HEAD_PREC=0 BENCH_ONLY=final timeout 300 python tools/bench_step_convs.py 2>&1 | tail -3
bash tools/gpu_pmc_conv.sh "D final 32->3 k7" "k_conv_head" head7 > /dev/null 2>&1
awk '{print $2, $3}' gpurun_out/pmc_conv_head7.txt
