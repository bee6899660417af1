#!/bin/bash
# round 6, call I: under-filled k_conv_hx tiles with the weight fragments straight into registers (template parameter BG; CADDY_HX_BG=0 restores the LDS-staged weight tiles):
# kernel parity, isolated shapes, step A/B (+ CADDY_MASK_FROM_X), Breakout-160, roll-out
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or hx or pre_split or lstm or register" > gpurun_out/i_kernels.txt 2>&1; tail -3 gpurun_out/i_kernels.txt
for v in 0 7; do echo "CADDY_HX_BG=$v"; CADDY_HX_BG=$v BENCH_ONLY="${1:-R }" timeout 600 python tools/bench_step_convs.py 2>&1 | grep -v amdgpu.ids; CADDY_HX_BG=$v BENCH_ONLY="E " timeout 600 python tools/bench_step_convs.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/i_shapes.txt; cat gpurun_out/i_shapes.txt
bash tools/gpu_ab.sh "LDS weight tiles|CADDY_HX_BG=0" "register weights|" "LDS weight tiles|CADDY_HX_BG=0" "register weights|" "mask from out|CADDY_MASK_FROM_X=0" "register weights|" > /dev/null
cat gpurun_out/ab.txt
for v in 0 7; do CADDY_HX_BG=$v timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('breakout160 hx_bg=$v full', d['ms_per_step'], 'erad', d['erad_only']['ms_per_step'])"; done
for v in 0 7; do echo "roll-out CADDY_HX_BG=$v"; CADDY_HX_BG=$v timeout 300 python tools/bench_rollout.py 2>&1 | grep -v amdgpu.ids | tail -4; done
