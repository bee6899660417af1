#!/bin/bash
# round 6, call E: tap row per barrier in the under-filled 64-channel k_conv_hx tiles (CADDY_HX_TAP_ROWS=0 restores one tap per barrier): kernel parity, isolated shapes, step A/B
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or hx" > gpurun_out/e_kernels.txt 2>&1; tail -3 gpurun_out/e_kernels.txt
for v in 0 1; do echo "CADDY_HX_TAP_ROWS=$v"; CADDY_HX_TAP_ROWS=$v BENCH_ONLY="${1:-R }" timeout 600 python tools/bench_step_convs.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/e_shapes.txt; cat gpurun_out/e_shapes.txt
bash tools/gpu_ab.sh "one tap|CADDY_HX_TAP_ROWS=0" "tap rows|" "one tap|CADDY_HX_TAP_ROWS=0" "tap rows|" "one tap|CADDY_HX_TAP_ROWS=0" "tap rows|" > /dev/null
cat gpurun_out/ab.txt
for v in 0 1; do CADDY_HX_TAP_ROWS=$v timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('breakout160 tap_rows=$v full', d['ms_per_step'], 'erad', d['erad_only']['ms_per_step'])"; done
