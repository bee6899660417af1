#!/bin/bash
# round 6, call F: K split of the accumulating split-bf16 launches with 200..256+ workgroups (CADDY_HX_SPLIT_BWD / CADDY_HX_SPLIT_TARGET) -> gpurun_out/ab.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "" "CADDY_HX_SPLIT_BWD=300" "CADDY_HX_SPLIT_BWD=300 CADDY_HX_SPLIT_TARGET=768"; do echo "[$v]"; env $v BENCH_ONLY="dgrad" timeout 600 python tools/bench_step_convs.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/f_shapes.txt; cat gpurun_out/f_shapes.txt
bash tools/gpu_ab.sh "base|" "split<300|CADDY_HX_SPLIT_BWD=300" "split<300 target 768|CADDY_HX_SPLIT_BWD=300 CADDY_HX_SPLIT_TARGET=768" "target 768|CADDY_HX_SPLIT_TARGET=768" "base|" "split<300|CADDY_HX_SPLIT_BWD=300" "split<300 target 768|CADDY_HX_SPLIT_BWD=300 CADDY_HX_SPLIT_TARGET=768" "target 768|CADDY_HX_SPLIT_TARGET=768" > /dev/null
cat gpurun_out/ab.txt
