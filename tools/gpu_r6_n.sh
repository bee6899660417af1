#!/bin/bash
# round 6, call N: chunked perceptual pass with the size threshold: parity tests, the three bench workloads (quick), one-call A/B on BAIR
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_host_api_gpu.py -m gpu -q -x -k "perceptual or deterministic or bair256 or trainer or plugin or evaluator" > gpurun_out/n_model.txt 2>&1; tail -3 gpurun_out/n_model.txt
for wl in breakout160_t9_b8 breakout64_t8_b4; do for v in 1 2; do CADDY_PERC_CHUNKS=$v timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl chunks_cfg=$v full', round(d['ms_per_step'],2), 'erad', round(d['erad_only']['ms_per_step'],2))"; done; done
bash tools/gpu_ab.sh "one pass|CADDY_PERC_CHUNKS=1" "default|" "one pass|CADDY_PERC_CHUNKS=1" "default|" "round-5 everything|CADDY_PERC_CHUNKS=1 CADDY_HX_BG=0 CADDY_MASK_FROM_X=0 CADDY_S16_GRADS=0" "default|" > /dev/null
cat gpurun_out/ab.txt
