cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx" > gpurun_out/pytest_hx.txt 2>&1; tail -5 gpurun_out/pytest_hx.txt
timeout 600 python tools/bench_hx.py > gpurun_out/bench_hx.txt 2>&1; cat gpurun_out/bench_hx.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2b_perc.json 2> gpurun_out/bench_r2b_perc.err; tail -c 400 gpurun_out/bench_r2b_perc.err
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-perceptual --no-rollout > gpurun_out/bench_r2b_noperc.json 2> gpurun_out/bench_r2b_noperc.err; tail -c 200 gpurun_out/bench_r2b_noperc.err
