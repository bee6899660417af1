#!/bin/bash
# round 6, call W: chunk count with the cut-once smaller levels, repeated
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_ab.sh "one pass|CADDY_PERC_CHUNKS=1" "2 chunks|CADDY_PERC_CHUNKS=2" "3 chunks|CADDY_PERC_CHUNKS=3" "4 chunks|CADDY_PERC_CHUNKS=4" "3 chunks|CADDY_PERC_CHUNKS=3" "4 chunks|CADDY_PERC_CHUNKS=4" "3 chunks|CADDY_PERC_CHUNKS=3" "2 chunks|CADDY_PERC_CHUNKS=2" "one pass|CADDY_PERC_CHUNKS=1" > /dev/null
cat gpurun_out/ab.txt
for v in -2 -3 -4 -3; do CADDY_PERC_CHUNKS=$v timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('breakout160 chunks=$v full', round(d['ms_per_step'],2))"; done
