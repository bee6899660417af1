cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe > gpurun_out/tr_probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2a_perc.json 2> gpurun_out/bench_r2a_perc.err; tail -c 600 gpurun_out/bench_r2a_perc.err
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-perceptual --no-rollout > gpurun_out/bench_r2a_noperc.json 2> gpurun_out/bench_r2a_noperc.err; tail -c 300 gpurun_out/bench_r2a_noperc.err
