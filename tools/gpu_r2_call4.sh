cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "perceptual or rollout or bair256" > gpurun_out/pytest_sel.txt 2>&1; tail -4 gpurun_out/pytest_sel.txt
rm -rf gpurun_out/prof_serial
CADDY_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout > gpurun_out/bench_serial.json 2> gpurun_out/bench_serial.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 45 > gpurun_out/step_breakdown_serial_perc.txt; cat gpurun_out/step_breakdown_serial_perc.txt
rm -rf gpurun_out/prof_serial
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; tail -c 300 gpurun_out/bench_r2d.err
CADDY_ROLLOUT_GRAPH=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_r2d_nograph.json 2> gpurun_out/bench_r2d_nograph.err; tail -c 100 gpurun_out/bench_r2d_nograph.err
