#!/bin/bash
# round 6, call Q: stream timeline + critical path of the traced FULL step (VGG19 perceptual pass in chunks beside the BPTT replay) and of the one-pass form
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 3 1; do
  rm -rf gpurun_out/prof_tl
  CADDY_PERC_CHUNKS=$v timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tl -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/bench_tl.err
  python tools/critical_path.py gpurun_out/prof_tl/bair_results.db 4.0 200.0 > gpurun_out/critical_path_full_chunks$v.txt 2>&1; head -40 gpurun_out/critical_path_full_chunks$v.txt | cut -c1-150
done
rm -rf gpurun_out/prof_tl
