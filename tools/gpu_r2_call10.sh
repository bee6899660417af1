bash tools/gpu_check.sh
cd ${GRAFT_REPO_ROOT:-.}
bash tools/gpu_pmc.sh bair256_t16_b8
timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-rollout > gpurun_out/bench_breakout160.json 2> gpurun_out/bench_breakout160.err; tail -c 200 gpurun_out/bench_breakout160.err
bash tools/gpu_pmc.sh breakout160_t9_b8
timeout 300 python bench.py --workload breakout64_t8_b4 --steps 20 --warmup 3 --no-rollout > gpurun_out/bench_breakout64.json 2> gpurun_out/bench_breakout64.err; tail -c 200 gpurun_out/bench_breakout64.err
