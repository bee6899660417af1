#!/bin/bash
# Full GPU verification pass (run through gpurun): parity tests, default bench, rocprofv3 kernel stats -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -4 gpurun_out/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.json
rm -rf gpurun_out/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python tools/rocprof_summary.py gpurun_out/prof_stats/bair_results.db "python bench.py --steps 3 --warmup 1 --no-cpu-baseline (BAIR 256x256, T=16, B=8; 4 steps + 1 profiled step + 32-frame roll-out)" > gpurun_out/kernel_stats.txt
head -8 gpurun_out/kernel_stats.txt
