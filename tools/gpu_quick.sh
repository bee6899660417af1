cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_host_api_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 --no-perceptual 2>&1 | grep "timed region"
