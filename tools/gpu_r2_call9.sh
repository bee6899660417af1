cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
rm -rf gpurun_out/pmc_hx
BENCH_ONLY="VGG 512->512 @32" timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_hx -o hx -- python tools/bench_hx.py > gpurun_out/pmc_hx.log 2>&1
python - <<'PY'
import sqlite3, collections
c = sqlite3.connect("gpurun_out/pmc_hx/hx_results.db")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name[:75]; agg[k][cn] += val; cnt[(k, cn)] += 1
for k, d in agg.items():
    if "conv_hxIDF16_Li2" not in k: continue
    for cn, v in sorted(d.items()): print(f"{k[-40:]}   {cn:32s} {v / cnt[(k, cn)]:16.0f} per launch")
PY
done
rm -rf gpurun_out/pmc_hx
