cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_rollout.py 64 2>&1 | grep run
CADDY_ROLLOUT_GRAPH=0 python tools/bench_rollout.py 64 2>&1 | grep run | sed 's/^/nograph /'
timeout 600 python tools/vgg_precision_study.py 2>&1 | grep -v "^$" | tail -8
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "rollout or full_model_parity or perceptual" 2>&1 | tail -3
