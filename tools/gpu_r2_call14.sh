cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx" 2>&1 | tail -2
timeout 600 python tools/bench_hx.py 2>&1 | grep TF | cut -c1-125
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 --no-perceptual 2>&1 | grep "timed region"
