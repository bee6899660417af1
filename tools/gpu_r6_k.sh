#!/bin/bash
# round 6, call K: weight-stationary XCD map of k_conv_hx (CADDY_HX_XCD_MAP=1 forces the activation-stationary order of rounds 2-5) on top of the register-weights variants
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or hx or pre_split or lstm" > gpurun_out/k_kernels.txt 2>&1; tail -3 gpurun_out/k_kernels.txt
for v in "CADDY_HX_BG=0 CADDY_HX_XCD_MAP=1" "CADDY_HX_XCD_MAP=1" "CADDY_HX_BG=0" ""; do echo "== $v"; env $v BENCH_ONLY="R " timeout 600 python tools/bench_step_convs.py 2>&1 | grep -v amdgpu.ids | grep -v wgrad; done > gpurun_out/k_shapes.txt; cat gpurun_out/k_shapes.txt
bash tools/gpu_ab.sh "round-5 kernels|CADDY_HX_BG=0 CADDY_HX_XCD_MAP=1 CADDY_MASK_FROM_X=0" "all new|" "LDS tiles, auto map|CADDY_HX_BG=0" "all new|" "registers, map 1|CADDY_HX_XCD_MAP=1" "all new|" > /dev/null
cat gpurun_out/ab.txt
for v in "CADDY_HX_BG=0 CADDY_HX_XCD_MAP=1" ""; do env $v timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('breakout160 [$v] full', d['ms_per_step'], 'erad', d['erad_only']['ms_per_step'])"; done
for v in "CADDY_HX_BG=0 CADDY_HX_XCD_MAP=1" "CADDY_HX_XCD_MAP=1" ""; do echo "roll-out [$v]"; env $v timeout 300 python tools/bench_rollout.py 2>&1 | grep -v amdgpu.ids | tail -2; done
