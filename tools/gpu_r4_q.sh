#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or batchnorm_fused or conv_case or test_conv" 2>&1 | tail -3
BENCH_ONLY=wgrad timeout 600 python tools/bench_step_convs.py $PWD/tools/_ab/libcaddy_hip_head.so $PWD/playablevideogeneration_amd/csrc/libcaddy_hip.so 2>&1 | grep "^wgrad\|^kind"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin --no-perceptual"
for i in 1 2; do
for lib in tools/_ab/libcaddy_hip_head.so playablevideogeneration_amd/csrc/libcaddy_hip.so; do echo $lib; CADDY_HIP_LIB=$PWD/$lib timeout 600 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('erad step ms', d['ms_per_step'])"; done
done
