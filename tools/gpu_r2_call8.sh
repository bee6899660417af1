cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx" 2>&1 | tail -3
timeout 600 python tools/bench_hx.py 2>&1 | grep TF
BENCH_KIND=wgrad BENCH_WGRAD_PREC=17 timeout 300 python tools/bench_conv.py 2>&1 | grep wgrad | head -12
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout > gpurun_out/bench_r2f.json 2> gpurun_out/bench_r2f.err; tail -c 200 gpurun_out/bench_r2f.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-perceptual --no-rollout > gpurun_out/bench_r2f_noperc.json 2> gpurun_out/bench_r2f_noperc.err; tail -c 100 gpurun_out/bench_r2f_noperc.err
