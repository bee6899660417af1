cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.txt
export BENCH_KIND=fwd
BENCH_ONLY=fdgrad timeout 300 python tools/bench_conv.py 2>&1 | grep fwd >> gpurun_out/ab.txt
BENCH_ONLY=stem timeout 300 python tools/bench_conv.py 2>&1 | grep fwd >> gpurun_out/ab.txt
unset BENCH_KIND
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k conv 2>&1 | grep -E "passed|failed" >> gpurun_out/ab.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-rollout 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('step', d['ms_per_step'])" >> gpurun_out/ab.txt
cat gpurun_out/ab.txt
