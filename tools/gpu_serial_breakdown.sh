#!/bin/bash
# serialised (CADDY_STREAMS=0) per-kernel breakdown of the full step for a list of "label|ENV=.." specs -> gpurun_out/breakdown_<label>.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  rm -rf gpurun_out/prof_serial
  env $envs CADDY_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin ${BD_ARGS} > /dev/null 2> gpurun_out/bd.err
  python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 70 > gpurun_out/breakdown_$label.txt; head -3 gpurun_out/breakdown_$label.txt
  rm -rf gpurun_out/prof_serial
done
