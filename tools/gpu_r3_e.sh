#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "parity or tight or full_geometry or baseline_geometry_properties or breakout160 or trainer" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_host_api_gpu.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
echo "dstream=0"; CADDY_D_STREAM=0 timeout 300 $B 2>&1 | grep "timed region"
echo "dstream=1"; timeout 300 $B 2>&1 | grep "timed region"
done
B="python bench.py --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
echo "full dstream=0"; CADDY_D_STREAM=0 timeout 300 $B 2>&1 | grep "timed region"
echo "full dstream=1"; timeout 300 $B 2>&1 | grep "timed region"
