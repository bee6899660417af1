#!/bin/bash
# round 6, call A: new parity cases + 7x7 head kernel A/B + critical-path records + host enqueue times -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv" > gpurun_out/a_kernels.txt 2>&1; tail -3 gpurun_out/a_kernels.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "tennis_native or slope_decision or test_full_model_parity or rollout_parity or tennis_rollout" > gpurun_out/a_model.txt 2>&1; grep -E "passed|failed|Error|rel_l2|flips|worst" gpurun_out/a_model.txt | cut -c1-600 | head -40
bash tools/gpu_ab.sh "r5|CADDY_HIP_LIB=tools/_ab/libcaddy_r5.so" "new|" "r5|CADDY_HIP_LIB=tools/_ab/libcaddy_r5.so" "new|"
for l in tools/_ab/libcaddy_r5.so ""; do CADDY_HIP_LIB=$l timeout 300 python tools/bench_rollout.py 2>&1 | tail -3; done
for wl in bair256_t16_b8 breakout160_t9_b8 breakout64_t8_b4; do timeout 300 python tools/host_time.py $wl 2>&1 | tail -5; done > gpurun_out/host_time.txt; cat gpurun_out/host_time.txt
for wl in bair256_t16_b8 breakout160_t9_b8; do
  rm -rf gpurun_out/prof_cp
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_cp -o cp -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/cp.err
  python tools/critical_path.py gpurun_out/prof_cp/cp_results.db 2.0 25 > gpurun_out/critical_path_${wl}_erad.txt 2>&1; head -3 gpurun_out/critical_path_${wl}_erad.txt | cut -c1-300
done
rm -rf gpurun_out/prof_cp
timeout 300 python tools/layer_profile.py > gpurun_out/layer_profile.txt 2>&1; grep -E "phase|k7" gpurun_out/layer_profile.txt
