#!/bin/bash
# refresh of the default bench line + roll-out kernel trace after the roll-out change
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; grep -E "timed region|roll-out|plugin|erad" gpurun_out/bench_default.err
bash tools/gpu_rollout_profile.sh 2>&1 | head -3
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
