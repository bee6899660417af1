cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx" 2>&1 | tail -3
timeout 600 python tools/bench_hx.py 2>&1 | grep TF
echo "== forced small variant"; CADDY_HX_BIG=0 BENCH_ONLY="VGG" timeout 600 python tools/bench_hx.py 2>&1 | grep TF
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err; tail -c 200 gpurun_out/bench_r2e.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-perceptual --no-rollout > gpurun_out/bench_r2e_noperc.json 2> gpurun_out/bench_r2e_noperc.err; tail -c 100 gpurun_out/bench_r2e_noperc.err
