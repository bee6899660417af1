#!/bin/bash
# round 6, call R: chunked perceptual pass forced on the smaller workloads (CADDY_PERC_CHUNKS=-n ignores the size threshold)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in breakout160_t9_b8 breakout64_t8_b4; do for v in 1 -2 -3 1 -2; do CADDY_PERC_CHUNKS=$v timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl chunks=$v full', round(d['ms_per_step'],2), 'erad', round(d['erad_only']['ms_per_step'],2))"; done; done
