#!/bin/bash
# round 6, call B: main-stream kernel sequence of the traced E/R/A/D step (BAIR), fixed parity cases, Adam zero-fill golden on the GPU -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "tennis_native or slope_decision or config_branches" > gpurun_out/b_model.txt 2>&1; grep -E "passed|failed|Error|rel_l2|flips|worst" gpurun_out/b_model.txt | cut -c1-400 | head -30
timeout 900 python -m pytest tests/test_host_api_gpu.py -m gpu -q -x -k "ensemble or zero_fill" > gpurun_out/b_host.txt 2>&1; tail -3 gpurun_out/b_host.txt
rm -rf gpurun_out/prof_cp
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_cp -o cp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/cp.err
python tools/chain_sequence.py gpurun_out/prof_cp/cp_results.db > gpurun_out/chain_bair_erad.txt 2>&1; head -2 gpurun_out/chain_bair_erad.txt
rm -rf gpurun_out/prof_cp
CADDY_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_cp -o cp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/cp.err
python tools/chain_sequence.py gpurun_out/prof_cp/cp_results.db > gpurun_out/chain_bair_erad_serial.txt 2>&1; head -2 gpurun_out/chain_bair_erad_serial.txt
rm -rf gpurun_out/prof_cp
