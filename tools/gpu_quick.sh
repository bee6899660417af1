# quick GPU check after a kernel change: the hx / perceptual tests, roll-out rate and the step time with and without the perceptual term
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_host_api_gpu.py -m gpu -x -q -k "perceptual or trainer or perc" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
CADDY_VGG_LEVELS_PARALLEL=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
done
