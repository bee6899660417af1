cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "folded or hx" 2>&1 | tail -2
python tools/bench_tiny.py 2>&1 | grep us
BENCH_ONLY="VGG 512->512 @32" python tools/bench_hx.py 2>&1 | grep TF | cut -c1-125
BENCH_ONLY="R lstm" python tools/bench_hx.py 2>&1 | grep TF | cut -c1-125
CADDY_ROLLOUT_GRAPH=0 python tools/bench_rollout.py 36 2>&1 | tail -1
python tools/bench_rollout.py 36 2>&1 | tail -1
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "rollout" 2>&1 | tail -2
