cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx" > gpurun_out/pytest_hx.txt 2>&1; tail -5 gpurun_out/pytest_hx.txt
BENCH_KIND=wgrad timeout 300 python tools/bench_conv.py > gpurun_out/bench_wgrad_fp32.txt 2>&1
BENCH_KIND=wgrad BENCH_WGRAD_PREC=17 timeout 300 python tools/bench_conv.py > gpurun_out/bench_wgrad_hx.txt 2>&1; paste -d'|' <(grep wgrad gpurun_out/bench_wgrad_fp32.txt | cut -c1-70) <(grep wgrad gpurun_out/bench_wgrad_hx.txt | cut -c30-70) | head -20
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2c_perc.json 2> gpurun_out/bench_r2c_perc.err; tail -c 300 gpurun_out/bench_r2c_perc.err
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-perceptual --no-rollout > gpurun_out/bench_r2c_noperc.json 2> gpurun_out/bench_r2c_noperc.err; tail -c 200 gpurun_out/bench_r2c_noperc.err
