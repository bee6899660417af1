cd $GRAFT_REPO_ROOT
BENCH_ONLY=final timeout 300 python tools/bench_conv.py 2>&1 | grep -E "fwd|wgrad" > gpurun_out/thin_ab.txt
BENCH_ONLY=dgrad timeout 300 python tools/bench_conv.py 2>&1 | grep -E "fdgrad|sdgrad" >> gpurun_out/thin_ab.txt
BENCH_ONLY=stem timeout 300 python tools/bench_conv.py 2>&1 | grep -E "stem" >> gpurun_out/thin_ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k conv 2>&1 | tail -2 >> gpurun_out/thin_ab.txt
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-rollout 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('step', d['ms_per_step'])" >> gpurun_out/thin_ab.txt
cat gpurun_out/thin_ab.txt
