#!/bin/bash
# round 6, call C: k_wgrad_hx halo-row-major MFMA phase -- kernel parity, isolated wgrad shapes (r5 library vs new), step A/B -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wgrad or conv" > gpurun_out/c_kernels.txt 2>&1; tail -3 gpurun_out/c_kernels.txt
BENCH_ONLY=wgrad timeout 600 python tools/bench_step_convs.py tools/_ab/libcaddy_r5.so "" > gpurun_out/c_wgrad_shapes.txt 2>&1; cat gpurun_out/c_wgrad_shapes.txt | tail -30
bash tools/gpu_ab.sh "r5|CADDY_HIP_LIB=tools/_ab/libcaddy_r5.so" "new|" "r5|CADDY_HIP_LIB=tools/_ab/libcaddy_r5.so" "new|" "r5|CADDY_HIP_LIB=tools/_ab/libcaddy_r5.so" "new|"
