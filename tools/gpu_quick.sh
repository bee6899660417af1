# quick GPU check after a kernel change: the hx / perceptual tests, roll-out rate and the step time with and without the perceptual term
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "hx or folded or perceptual or pool" 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_rollout.py 36 2>&1 | tail -1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
CADDY_VGG_FUSE_POOL=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 --no-perceptual 2>&1 | grep "timed region"
