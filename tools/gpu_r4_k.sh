#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "test_conv" 2>&1 | tail -2
timeout 300 python tools/layer_profile.py > gpurun_out/lp_new.txt 2>&1
grep "Cout=    3" gpurun_out/lp_new.txt | grep fwd
grep "phase fwd:closed\|phase fwd:teacher" gpurun_out/lp_new.txt
B="python bench.py --no-perceptual --no-cpu-baseline --no-rollout --no-extra-legs --profile-steps 0 --steps 10 --warmup 3"
for i in 1 2; do
  echo "base lib:"; CADDY_HIP_LIB=$PWD/tools/_ab/libcaddy_hip_base.so timeout 300 $B 2>&1 | grep "timed region"
  echo "this tree:"; timeout 300 $B 2>&1 | grep "timed region"
done
