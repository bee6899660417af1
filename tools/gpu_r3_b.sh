#!/bin/bash
# A/B: why did the roll-out rate drop inside the new bench.py? (a) stand-alone roll-out, (b) bench without the plugin leg, (c) bench with it
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bench_rollout.py 2>&1 | grep run
timeout 600 python bench.py --no-cpu-baseline --no-plugin 2>&1 >/dev/null | grep -E "roll-out|timed"
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs 2>&1 >/dev/null | grep -E "roll-out|timed"
timeout 600 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep -E "roll-out|plugin"
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
