#!/bin/bash
# roll-out: frame graph launched on the caller's stream (1, default) vs eager launches (0) vs the graph on the internal stream (2, the round-2 form); parity first
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_host_api_gpu.py -m gpu -x -q -k "rollout or sampler or drivers or plugin" 2>&1 | tail -3
for i in 1 2; do for v in 1 0 2; do echo "CADDY_ROLLOUT_GRAPH=$v"; CADDY_ROLLOUT_GRAPH=$v timeout 300 python tools/bench_rollout.py 36 2>&1 | grep "run"; done; done
