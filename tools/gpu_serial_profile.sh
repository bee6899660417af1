#!/bin/bash
# per-kernel cost without stream sharing: serialised step under rocprofv3 -> gpurun_out/prof_serial/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_serial
CADDY_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_serial.json 2> gpurun_out/bench_serial.err
