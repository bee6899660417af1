#!/bin/bash
# HBM-traffic counters (separate passes, --kernel-trace only, as the MI355X guide prescribes) -> gpurun_out/pmc_traffic_<workload>[_erad].json
#   bash tools/gpu_pmc.sh [workload] [erad]      (default bair256_t16_b8; "erad": the E/R/A/D-only step, --no-perceptual)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
WL=${1:-bair256_t16_b8}
SFX=""; EXTRA=""; PERC=1
if [ "$2" = "erad" ]; then SFX="_erad"; EXTRA="--no-perceptual"; PERC=0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -o run -- python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs $EXTRA > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/run_results.db gpurun_out/pmc_WRITE_SIZE/run_results.db $WL $PERC 2 > gpurun_out/pmc_traffic_$WL$SFX.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -c 300 gpurun_out/pmc_traffic_$WL$SFX.json
