#!/bin/bash
# round 6, call V: half- and quarter-resolution levels of the perceptual pass cut once (late half with the first chunk, early half in front of chunk nch / 2)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "perceptual or deterministic or bair256" > gpurun_out/v_model.txt 2>&1; tail -3 gpurun_out/v_model.txt
bash tools/gpu_ab.sh "one pass|CADDY_PERC_CHUNKS=1" "4 chunks|" "5 chunks|CADDY_PERC_CHUNKS=5" "6 chunks|CADDY_PERC_CHUNKS=6" "3 chunks|CADDY_PERC_CHUNKS=3" "one pass|CADDY_PERC_CHUNKS=1" "4 chunks|" "5 chunks|CADDY_PERC_CHUNKS=5" "6 chunks|CADDY_PERC_CHUNKS=6" > /dev/null
cat gpurun_out/ab.txt
for v in 1 -3 -4 1 -4; do CADDY_PERC_CHUNKS=$v timeout 300 python bench.py --workload breakout160_t9_b8 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('breakout160 chunks=$v full', round(d['ms_per_step'],2), 'erad', round(d['erad_only']['ms_per_step'],2))"; done
