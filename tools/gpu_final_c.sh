#!/bin/bash
# Round-end refresh after late changes: full GPU suite, smoke, the default bench line and the two Breakout lines -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; grep -E "timed region|erad_only|exact|roll-out|plugin" gpurun_out/bench_default.err
timeout 600 python bench.py --workload breakout160_t9_b8 --no-rollout --no-plugin > gpurun_out/bench_breakout160.json 2> gpurun_out/bench_breakout160.err; grep -E "timed region" gpurun_out/bench_breakout160.err
timeout 600 python bench.py --workload breakout64_t8_b4 --no-rollout --no-plugin > gpurun_out/bench_breakout64.json 2> gpurun_out/bench_breakout64.err; grep -E "timed region" gpurun_out/bench_breakout64.err
