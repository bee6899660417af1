#!/bin/bash
# round 6, call G: pre-split (S16-bf16) conv-output gradients: kernel + model parity, step A/B (CADDY_S16_GRADS=0 = fp32 exchange), isolated shapes
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "pre_split or batchnorm or lstm or pool" > gpurun_out/g_kernels.txt 2>&1; tail -5 gpurun_out/g_kernels.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "pre_split or test_full_model_parity or deterministic" > gpurun_out/g_model.txt 2>&1; grep -E "passed|failed|Error|pre_split_gradient|assert" gpurun_out/g_model.txt | cut -c1-300 | head -20
bash tools/gpu_ab.sh "fp32 dY|CADDY_S16_GRADS=0" "S16 dY|" "fp32 dY|CADDY_S16_GRADS=0" "S16 dY|" "fp32 dY|CADDY_S16_GRADS=0" "S16 dY|" > /dev/null
cat gpurun_out/ab.txt
for v in 0 1; do CADDY_S16_GRADS=$v timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('breakout160 s16_grads=$v full', d['ms_per_step'], 'erad', d['erad_only']['ms_per_step'])"; done
