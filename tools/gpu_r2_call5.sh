cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_rollout.py 64 2>&1 | grep run
CADDY_ROLLOUT_GRAPH=0 python tools/bench_rollout.py 64 2>&1 | grep run | sed 's/^/nograph /'
rm -rf gpurun_out/prof_roll
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_roll -o roll -- python tools/bench_rollout.py 32 > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_roll/roll_results.db "rollout" | head -40 > gpurun_out/rollout_kernels.txt; cat gpurun_out/rollout_kernels.txt | cut -c1-80,110-160
rm -rf gpurun_out/prof_roll gpurun_out/pmc_hx
BENCH_ONLY="VGG 512->512 @32" timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_hx -o hx -- python tools/bench_hx.py > gpurun_out/pmc_hx.log 2>&1
python - <<'PY'
import sqlite3, collections
c = sqlite3.connect("gpurun_out/pmc_hx/hx_results.db")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name[:70]; agg[k][cn] += val; cnt[(k, cn)] += 1
for k, d in agg.items():
    if "conv_hx" not in k: continue
    print(k)
    for cn, v in sorted(d.items()): print(f"   {cn:28s} {v / cnt[(k, cn)]:14.0f} per launch")
PY
rm -rf gpurun_out/pmc_hx
