#!/bin/bash
# Round-end measurement pass, part A (run through gpurun): parity tests, smoke, PMC traffic of this round (full step + E/R/A/D-only step), the bench lines -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
bash tools/gpu_pmc.sh bair256_t16_b8 > /dev/null 2>&1
cp gpurun_out/pmc_traffic_bair256_t16_b8.json profiles/r05_pmc_traffic_bair256_t16_b8.json
bash tools/gpu_pmc.sh bair256_t16_b8 erad > /dev/null 2>&1
cp gpurun_out/pmc_traffic_bair256_t16_b8_erad.json profiles/r05_pmc_traffic_bair256_t16_b8_erad.json
bash tools/gpu_pmc.sh breakout160_t9_b8 > /dev/null 2>&1
cp gpurun_out/pmc_traffic_breakout160_t9_b8.json profiles/r05_pmc_traffic_breakout160_t9_b8.json
bash tools/gpu_pmc.sh breakout160_t9_b8 erad > /dev/null 2>&1
cp gpurun_out/pmc_traffic_breakout160_t9_b8_erad.json profiles/r05_pmc_traffic_breakout160_t9_b8_erad.json
cp profiles/r05_pmc_traffic_*.json gpurun_out/
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
timeout 600 python bench.py --workload breakout160_t9_b8 --no-rollout --no-plugin > gpurun_out/bench_breakout160.json 2> gpurun_out/bench_breakout160.err; tail -c 200 gpurun_out/bench_breakout160.err
timeout 600 python bench.py --workload breakout64_t8_b4 --no-rollout --no-plugin > gpurun_out/bench_breakout64.json 2> gpurun_out/bench_breakout64.err; tail -c 200 gpurun_out/bench_breakout64.err
