#!/bin/bash
# Round-6 measurement pass, part A (through gpurun): full GPU suite, smoke, the bench lines, PMC traffic (full step, E/R/A/D-only step, Breakout-160 E/R/A/D) -> gpurun_out/, profiles/r06_*
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; grep -E "timed region|erad_only|exact|roll-out|plugin" gpurun_out/bench_default.err
timeout 600 python bench.py --workload breakout160_t9_b8 --no-rollout --no-plugin > gpurun_out/bench_breakout160.json 2> gpurun_out/bench_breakout160.err; grep -E "timed region|erad_only" gpurun_out/bench_breakout160.err
timeout 600 python bench.py --workload breakout64_t8_b4 --no-rollout --no-plugin > gpurun_out/bench_breakout64.json 2> gpurun_out/bench_breakout64.err; grep -E "timed region|erad_only" gpurun_out/bench_breakout64.err
bash tools/gpu_pmc.sh bair256_t16_b8 > /dev/null 2>&1
bash tools/gpu_pmc.sh bair256_t16_b8 erad > /dev/null 2>&1
bash tools/gpu_pmc.sh breakout160_t9_b8 erad > /dev/null 2>&1
ls -la gpurun_out/pmc_traffic_*.json
