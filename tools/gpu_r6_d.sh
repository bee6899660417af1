#!/bin/bash
# round 6, call D: step A/B of the halo-row-major k_wgrad_hx (base = HEAD's kernel in tools/_ab/libcaddy_base.so) -> gpurun_out/ab.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_ab.sh "base|CADDY_HIP_LIB=tools/_ab/libcaddy_base.so" "new|" "base|CADDY_HIP_LIB=tools/_ab/libcaddy_base.so" "new|" "base|CADDY_HIP_LIB=tools/_ab/libcaddy_base.so" "new|" > /dev/null
cat gpurun_out/ab.txt
