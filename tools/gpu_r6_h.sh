#!/bin/bash
# round 6, call H: issued work per kernel family of the E/R/A/D step (SQ counters, separate passes, --kernel-trace only) + the fixed lstm producer test
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "pre_split" > gpurun_out/h_kernels.txt 2>&1; tail -3 gpurun_out/h_kernels.txt
WL=${1:-bair256_t16_b8}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_w$i
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_w$i -o run -- python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-perceptual > gpurun_out/pmc_w$i.log 2>&1 || tail -3 gpurun_out/pmc_w$i.log
done
python tools/pmc_by_kernel.py 2 gpurun_out/pmc_w1/run_results.db gpurun_out/pmc_w2/run_results.db gpurun_out/pmc_w3/run_results.db > gpurun_out/issued_work_${WL}_erad.txt 2>&1
rm -rf gpurun_out/pmc_w1 gpurun_out/pmc_w2 gpurun_out/pmc_w3
head -40 gpurun_out/issued_work_${WL}_erad.txt | cut -c1-180
