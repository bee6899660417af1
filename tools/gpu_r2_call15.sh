cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "folded or hx" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "rollout" 2>&1 | tail -3
python tools/rollout_layer_profile.py 2>&1 | tail -40
CADDY_ROLLOUT_GRAPH=1 python tools/bench_rollout.py 36 2>&1 | tail -3
CADDY_ROLLOUT_FOLD=0 python tools/bench_rollout.py 36 2>&1 | tail -3
