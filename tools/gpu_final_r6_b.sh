#!/bin/bash
# Round-6 measurement pass, part B: rocprofv3 kernel stats of the TIMED STEP ONLY (no extra legs: the top rows are the timed region's), serialised breakdowns (full and E/R/A/D-only),
# stream timeline + critical path of the E/R/A/D step, roll-out kernel trace, per-layer profile -> gpurun_out/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats
CMD="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o bair -- $CMD > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python tools/rocprof_summary.py gpurun_out/prof_stats/bair_results.db "$CMD (BAIR 256x256, T=16, B=8 incl. VGG19 perceptual loss: 1 warm-up + 4 timed steps, nothing else)" > gpurun_out/kernel_stats.txt 2>&1
head -12 gpurun_out/kernel_stats.txt | cut -c1-90,105-160
rm -rf gpurun_out/prof_stats gpurun_out/prof_serial
CADDY_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/bench_serial.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 60 > gpurun_out/step_breakdown_serial.txt; head -3 gpurun_out/step_breakdown_serial.txt
rm -rf gpurun_out/prof_serial
CADDY_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/bench_serial_erad.err
python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db 60 > gpurun_out/step_breakdown_serial_erad.txt; head -3 gpurun_out/step_breakdown_serial_erad.txt
rm -rf gpurun_out/prof_serial gpurun_out/prof_tl
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tl -o bair -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-rollout --no-perceptual --no-extra-legs --no-plugin > /dev/null 2> gpurun_out/bench_tl.err
python tools/timeline.py gpurun_out/prof_tl/bair_results.db 40 > gpurun_out/timeline_erad.txt 2>&1; sed -n 2,7p gpurun_out/timeline_erad.txt
python tools/critical_path.py gpurun_out/prof_tl/bair_results.db > gpurun_out/critical_path_erad.txt 2>&1; head -2 gpurun_out/critical_path_erad.txt
rm -rf gpurun_out/prof_tl
bash tools/gpu_pmc.sh breakout160_t9_b8 > /dev/null 2>&1
bash tools/gpu_rollout_profile.sh 2>&1 | head -3
timeout 300 python tools/layer_profile.py > gpurun_out/layer_profile.txt 2>&1; grep phase gpurun_out/layer_profile.txt
